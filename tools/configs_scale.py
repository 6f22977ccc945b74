"""The `configs` block of bench.py (tools/bench_configs.py: C2b, C3, C5-literal, C5 + dense error model; C4 has its own driver,
tools/c4_scale.py) under N ranks, one per GPU, weak scaling: every rank runs the full per-GPU chain count on its device with its
chains keyed by global id, no data-path collective; rank 0 prints ONE JSON line per configuration with the whole-job rate (sum of the
ranks' evaluations over the slowest rank's time) and every rank's own rate.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/configs_scale.py

TINYDA_BENCH_ONE_GPU=1: every rank on cuda:0 over gloo (rehearsal of the code path, not a measurement)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402

from tinyda_amd import distributed as tdist  # noqa: E402
import tools.bench_configs as bc  # noqa: E402


def main():
    small = "--small" in sys.argv
    one_gpu = os.environ.get("TINYDA_BENCH_ONE_GPU") == "1"
    rank, local_rank, world = tdist.init_process_group("gloo" if one_gpu else None)
    if world > torch.cuda.device_count() and not one_gpu:
        raise SystemExit("%d ranks but %d device(s): refusing (TINYDA_BENCH_ONE_GPU=1 rehearses on one GPU)" % (world, torch.cuda.device_count()))
    devi = 0 if one_gpu else local_rank
    torch.cuda.set_device(devi)
    dev = torch.device("cuda", devi)
    bc.PLACE.update(device=devi, rank=rank)
    if small:  # rehearsal sizes
        runs = (("C3", lambda: bc.run_hierarchy("C3 (rehearsal size)", (256, 2048), [10], dict(kind=1, scaling=0.02), 8, 58982.0, "k_da_steps", N=256)),
                ("C5+AEM", lambda: bc.run_c5_aem(N=128, m=32, n_fine=4)))
    else:
        runs = (("C2b", bc.run_c2b), ("C3", bc.run_c3), ("C5-literal", bc.run_c5), ("C5+AEM128", lambda: bc.run_c5_aem(m=128)))
    for tag, fn in runs:
        tdist.barrier()
        r = fn()
        secs = tdist.reduce_scalar(r["seconds"], "max", None if one_gpu else dev)
        evals = r["evals_per_s"] * r["seconds"]
        rates = [None] * world
        if world > 1:
            import torch.distributed as dist

            dist.all_gather_object(rates, r["evals_per_s"])
        else:
            rates = [r["evals_per_s"]]
        if rank == 0:
            print(json.dumps({"tag": tag, "name": r["name"], "n_gpus": world, "chains_per_gpu": r["chains"], "evals_per_s": evals * world / secs,
                              "per_rank_evals_per_s": rates, "seconds_max_over_ranks": secs,
                              "rccl_ranks": world if (world > 1 and not one_gpu) else (0 if world == 1 else "gloo rehearsal on one GPU"),
                              "scaling": "weak", "collectives_on_the_data_path": 0}), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
