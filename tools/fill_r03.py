"""Fill the R3_* placeholders of DESIGN.md's round-3 table from profiles/r03_bench.json (+ the cfg3 share run)."""
import json, re, sys
j = json.load(open('profiles/r03_bench.json'))
c3 = json.load(open('profiles/r03_bench_cfg3_share.json'))
r = j['roofline']; pr = j['path_roofline']
vals = {
    'R3_VALUE': f"{j['value']:,.0f}", 'R3_MS': f"{j['ms_per_step']:.2f}", 'R3_STREAMS': str(j['config']['hip_streams']),
    'R3_HBMM': f"{100 * pr['hbm_frac_vs_measured_peak']:.1f}", 'R3_HBM': f"{100 * pr['hbm_frac']:.1f}", 'R3_MFMA': f"{100 * pr['mfma_frac']:.1f}",
    'R3_BF16': f"{j['bf16_value']:,.0f}", 'R3_STAIN': f"{j['with_reinhard_value']:,.0f}", 'R3_FULL': f"{j['full_mode_value']:,.0f}",
    'R3_F32': f"{j['f32_value']:,.0f}", 'R3_CFG3': f"{c3['value']:,.0f}", 'R3_TFR': f"{j['tfrecords']['value']:,.0f}",
    'R3_DEC': f"{j['tfrecords']['decode_only_tiles_per_s']:,.0f}", 'R3_CPU': f"{j['cpu_baseline']['value']:.2f}",
    'R3_DOM': f"{r['avg_launch_ms']:.4f}", 'R3_TF': f"{r['achieved']:.0f}", 'R3_FRAC': f"{100 * r['frac']:.1f}",
}
for path in sys.argv[1:] or ['DESIGN.md']:
    s = open(path).read()
    for k in sorted(vals, key=len, reverse=True):
        s = s.replace(k, vals[k])
    open(path, 'w').write(s)
    print(path, 'left:', re.findall(r'R3_[A-Z0-9]+', s))
