"""Worker of tests/test_gpu_baseline_sizes.py::test_config4_n320_rollout_sharded_state_ranks_sharing_one_gpu.

usage: python _gpu_shared_rollout.py RANK WORLD PORT OUT WORKLOAD STEPS
BASELINE config 4 on the N > 1 route exactly as ``bench.py --rollout STEPS --gpus N`` steps it: ``sharded_forward`` with a
local output, ``advance_sharded_state`` between the lead times (one grid-halo all-to-all-v, no all-gather), one
``sharded_state_output`` at the end -- against the single-device chain ``model(x)`` + ``anemoi_advance_input`` on the same
weights and state.  Every rank runs the HIP kernels on cuda:0; the collectives are gloo staged through host memory.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out, workload, steps = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4],
                                               sys.argv[5], int(sys.argv[6]))
    os.environ["ANEMOI_AMD_DTYPE"] = "bf16"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from anemoi_models_amd import _lib

    _lib.load()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from anemoi_models_amd import ops
        from anemoi_models_amd.distributed.partition import (advance_sharded_state, sharded_forward,
                                                              sharded_state_output)

        device = torch.device("cuda", 0)
        model, graph, x, idx = bench.build(workload, device)  # seeded: the same weights and state on every rank
        cmap = torch.full((idx.num_input,), -1, dtype=torch.int32)
        cmap[idx.internal_model.input.prognostic] = idx.internal_model.output.prognostic.to(torch.int32)
        cmap = cmap.to(device)
        group = dist.group.WORLD
        with torch.no_grad():
            state = x.clone()
            for lead in range(steps):
                want = model(state)
                if lead + 1 < steps:
                    ops.advance_input(state, want, cmap)
            state = x.clone()
            for lead in range(steps):
                y_local, sp = sharded_forward(model, state, group, local_output=True)
                if lead + 1 < steps:
                    advance_sharded_state(model, state, y_local, sp, cmap)
            got = sharded_state_output(model, state, y_local, sp, group)
        torch.cuda.synchronize()
        info = dict(err=float((got.float() - want.float()).abs().max()), scale=float(want.abs().max()),
                    finite=bool(torch.isfinite(got).all()), shape=tuple(got.shape), own=sp.hi - sp.lo,
                    grid_halo=int(sp.grid_halo_ids.numel()), grid=int(x.shape[3]))
        torch.save(info, f"{out}.{rank}")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
