"""DESIGN.md and README.md from their templates (tools/templates/*.in: the text, with RND_* placeholders where a measured number goes),
profiles/<tag>_bench.json and profiles/<tag>_bench_cfg3_share.json: python tools/fill_design.py r06 <cpu tests> <gpu tests>.
Edit the TEMPLATES, then run this."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
d = json.load(open(f'profiles/{tag}_bench.json'))
ks = d['kernels']
rows = ['| kernel (profiling class) | file | launches | ms per launch | frac of its roofline |', '|---|---|---:|---:|---:|']
where = {'sepconv_k728_n728_19x19': 'kernels_wide.hip', 'sepconv_k728_n728_37x37': 'kernels_wide.hip', 'sepconv_k256_n728_37x37': 'kernels_wide.hip',
         'sepconv_k728_n1024_19x19': 'kernels_wide.hip (two launches of 512 columns)', 'blocktail_147_c128': 'kernels_stream.hip (block 2 tail, fused)',
         'blocktail_74_c256': 'kernels_stream.hip (block 3 tail, cooperative)', 'sepconv_k64_n128_147x147': 'kernels_stream.hip',
         'sepconv_k128_n256_74x74': 'kernels_stream.hip', 'front_stage_stem_conv2': 'kernels_front.hip (uint8 -> conv2, fused)',
         'respool_37_c728': 'kernels_split.hip (shortcut GEMM + pool + add)', 'respool_19_c1024': 'kernels_split.hip',
         'gemm_gap_sepconv_k1536_n2048_10x10': 'kernels_exit.hip (+ global average pool)', 'gemm_sepconv_k1024_n1536_10x10': 'kernels_exit.hip',
         'dw3x3_sepconv_k1536_n2048_10x10': 'kernels_split.hip (depthwise)', 'dw3x3_sepconv_k1024_n1536_10x10': 'kernels_split.hip (depthwise)',
         'mc_head_dense0': 'kernels_head.hip', 'mc_head_dense1': 'kernels_head.hip', 'mc_head_softmax_welford': 'kernels_misc.hip (head_final)',
         'stage_stats': 'kernels_misc.hip', 'slide_reduce': 'kernels_misc.hip'}
for k in ks:
    rows.append(f"| `{k['name']}` | {where.get(k['name'], '')} | {k['launches_per_step']:.0f} | {k['ms_per_launch']:.4f} | {k['frac_of_bound']:.2f} |")
t = d['tfrecords']
c3 = json.load(open(f'profiles/{tag}_bench_cfg3_share.json'))
wide = next(k for k in ks if k['name'] == d['roofline']['kernel'])
rep = {
    'RND_TAG': tag,
    'RND_MON_CHECKS': str(c3.get('range_monitor', {}).get('checks', '?')),
    'RND_MON_COST': (f"{100 * c3['range_monitor']['cost_frac']:+.1f} %" if c3.get('range_monitor') else '?'), 'RND_CORES': str(d['cpu_baseline']['cores']),
    'RND_CFG3_TABLE': f"{c3['with_table_value']:,.0f}", 'RND_CFG3_RATIO': f"{c3['tile_table']['ratio_to_value']:.3f}", 'RND_CFG3': f"{c3['value']:,.0f}",
    'RND_TFR_TABLE': f"{t['with_table_value'] / 1e3:.1f} k",
    'RND_WIDE_SHARE': f"{100 * d['roofline']['share_of_step']:.0f}",
    'RND_TRAFFIC': (f"{d['roofline']['traffic'] / 1e6:.0f}" if d['roofline'].get('traffic') else 'n/a'),
    'RND_VALUE': f"{d['value']:,.0f}", 'RND_MS': f"{d['ms_per_step']:.2f}",
    'RND_HBM': f"{100 * d['path_roofline']['hbm_frac']:.1f}", 'RND_MFMA': f"{100 * d['path_roofline']['mfma_frac']:.1f}",
    'RND_CPU_TESTS': sys.argv[2] if len(sys.argv) > 2 else '?', 'RND_GPU_TESTS': sys.argv[3] if len(sys.argv) > 3 else '?',
    'RND_CPU': f"{d['cpu_baseline']['value']:.2f}",
    'RND_WIDE_MS': f"{d['roofline']['avg_launch_ms']:.4f}", 'RND_WIDE_FRAC': f"{100 * d['roofline']['frac']:.1f} %",
    'RND_ENTRY': f"{d['entry_side_ms']:.2f}",
    'RND_KERNEL_TABLE': '\n'.join(rows),
    'RND_TFR_FRAC': f"{100 * max(t['value'], t['gpu_unfilter_value']) / d['value']:.0f} %",
    'RND_TFR': f"{t['value'] / 1e3:.1f} k tiles/s with the filters on the host, {t['gpu_unfilter_value'] / 1e3:.1f} k with them on the GPU (photo-like tiles: {t['photo']['value'] / 1e3:.1f} k / {t['photo']['gpu_unfilter_value'] / 1e3:.1f} k)",
    'RND_DEC': f"{t['decode_only_tiles_per_s'] / 1e3:.1f} k tiles/s ({t['decode_rows_only_tiles_per_s'] / 1e3:.1f} k stopping at the scanlines)",
    'RND_HOST': f"{d['host_tiles']['value'] / 1e3:.1f} k tiles/s",
}
s = open('tools/templates/DESIGN.md.in').read()
for k in sorted(rep, key=len, reverse=True):
    s = s.replace(k, rep[k])
open('DESIGN.md', 'w').write(s)
r = open('tools/templates/README.md.in').read()
line = (f"{d['value']:,.0f} tiles/s ({d['ms_per_step']:.2f} ms per batch; bf16 {d['bf16_value']:,.0f}), {d['with_reinhard_value']:,.0f} with the `reinhard_fast` "
        f"stain normaliser in the timed region, {d['full_mode_value']:,.0f} with the reference's loop structure (30 complete passes), {d['f32_value']:,.0f} on the "
        f"exact fp32 kernels, {d['host_tiles']['value'] / 1e3:.1f} k when the decoded tiles start in pageable host memory (the PCIe-inclusive rate), "
        f"{t['value'] / 1e3:.1f} k end to end from PNG TFRecords on {t['host_cores']} host cores ({t['gpu_unfilter_value'] / 1e3:.1f} k with the PNG scanline filters "
        f"reversed on the GPU), against {d['cpu_baseline']['value']:.2f} tiles/s for the PyTorch-CPU restatement on the same {d['cpu_baseline']['cores']} cores")
for k, v in (('RND_README_LINE', line), ('RND_TAG', tag), ('RND_CFG3_TABLE', rep['RND_CFG3_TABLE']), ('RND_CFG3_RATIO', rep['RND_CFG3_RATIO']),
             ('RND_CFG3', rep['RND_CFG3'])):
    r = r.replace(k, v)
open('README.md', 'w').write(r)
print('filled')
