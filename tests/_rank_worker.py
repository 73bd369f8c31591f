"""Child process of the multi-rank GPU tests: one rank of ``inference.evaluate`` with the real ``Engine``
(several ranks may share one GPU: argv[6] = local device, argv[7] = process-group backend, e.g. "0 gloo"; argv[8] = "group1":
a process group even for a world of one).  Writes its view of the result to ``argv[1]``.rank{r}.npz."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))


def build_slides(counts):
    from biscuit_amd.inference import Slide
    from biscuit_amd.synthetic import make_tiles
    return [Slide(f's{i}', make_tiles(c, seed=900 + i) if c else np.zeros((0, 299, 299, 3), np.uint8), c, y_true=i % 2)
            for i, c in enumerate(counts)]


def main():
    out, dtype, mc_n, batch = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    counts = [int(x) for x in sys.argv[5].split(',')]
    local_device = int(sys.argv[6]) if len(sys.argv) > 6 else None
    backend = sys.argv[7] if len(sys.argv) > 7 else None
    from biscuit_amd import distributed as D
    from biscuit_amd.engine import Engine
    from biscuit_amd.inference import evaluate
    from biscuit_amd.weights import synthetic_weights
    group1 = len(sys.argv) > 8 and sys.argv[8] == 'group1'
    rank, world, local = D.init_from_env('cuda', backend=backend, local_device=local_device, single_rank_group=group1)
    eng = Engine(synthetic_weights(1), dtype=dtype, max_batch=batch, max_mc=mc_n, device=local)
    # BQ_TEST_SAVE_DIR: also stream the tile table (this rank's shard; rank 0 splices the shards after the gather)
    res = evaluate(eng, build_slides(counts), mc_n=mc_n, seed=77, batch=batch, rank=rank, world=world,
                   save_dir=os.environ.get('BQ_TEST_SAVE_DIR'))
    np.savez(f'{out}.rank{rank}.npz', slide_pred=res.slide_pred, slide_unc=res.slide_unc, slide_count=res.slide_count,
             local=np.array(res.local_slides), tile_slide=np.array(res.tile_df['slide'], dtype=str),
             tile_pred=res.tile_df['cohort-y_pred1'].to_numpy(), tile_unc=res.tile_df['cohort-uncertainty1'].to_numpy())
    import torch.distributed as dist
    if dist.is_initialized():
        if group1:
            np.savez(f'{out}.group.npz', backend=dist.get_backend(), size=dist.get_world_size())
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
