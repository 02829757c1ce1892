"""Training entry point (reference main.py:4-26): Logger -> ddp_setup -> ExampleTrainer ->
train() -> destroy_process().  Launch under torchrun via run.sh, or plainly as
``python main.py --synthetic`` for a single process.

The reference hard-codes every setting as a Python literal; the same literals are the defaults
here and a thin CLI overlay makes the benchmark configs selectable.
"""
import argparse
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (multi-process RCCL on this pool)

import torch

from example_trainer import ExampleTrainer
from utils.logger import Logger


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--backend", default=None, help="nccl (RCCL) or gloo; default: nccl if a GPU is visible")
    ap.add_argument("--train-path", default="./data/train")
    ap.add_argument("--val-path", default="./data/val")
    ap.add_argument("--synthetic", action="store_true", help="use the deterministic synthetic dataset")
    ap.add_argument("--synthetic-size", type=int, default=1024)
    ap.add_argument("--labels", default="cat,dog,snake")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--batch-size", type=int, default=16, help="GLOBAL batch (split over ranks)")
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--save-period", type=int, default=5)
    ap.add_argument("--save-folder", default="./runs")
    ap.add_argument("--snapshot", default=None, help="resume from this snapshot (e.g. runs/weights/last.pth)")
    ap.add_argument("--no-validate", action="store_true")
    ap.add_argument("--batch-norm", action="store_true", help="VGG16-BN variant")
    ap.add_argument("--uint8-input", action="store_true",
                    help="ship uint8 pixels to the GPU; /255, mean and std are fused into the conv1 ingest kernel")
    ap.add_argument("--fp8-classifier", action="store_true", help="linear1/linear2 forward GEMMs in OCP e4m3 (GPU)")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--grad-accum", type=int, default=1)
    ap.add_argument("--num-workers", type=int, default=0)
    ap.add_argument("--steps", type=int, default=None, help="cap on steps per epoch (smoke runs)")
    ap.add_argument("--no-progress", action="store_true")
    ap.add_argument("--device-cache", action="store_true",
                    help="materialise the dataset once on the GPU and gather batches there (fits-in-HBM datasets)")
    ap.add_argument("--ring-loader", action="store_true",
                    help="synthetic training data from native producer threads through the pinned host ring "
                         "(uint8 over PCIe on the copy stream); use with --synthetic --no-validate")
    ap.add_argument("--hip-graph", action="store_true", help="capture the train step into one hipGraph (data-parallel: native RCCL transport only)")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="reduce-scatter + per-bucket sharded optimizer + all-gather instead of all-reduce")
    ap.add_argument("--no-bucket-optimizer", action="store_true",
                    help="world > 1: one optimizer pass after the last gradient bucket instead of per bucket "
                         "behind its all-reduce (the default)")
    ap.add_argument("--seed", type=int, default=0, help="seeds torch (init, CPU dropout) and keys the GPU dropout masks")
    ap.add_argument("--nproc", type=int, default=0,
                    help="self-launch this many local ranks (one per GPU) without torchrun; 0: run in this process")
    ap.add_argument("--max-restarts", type=int, default=0,
                    help="with --nproc: relaunch a failed job this many times; restarted jobs resume from last.pth")
    ap.add_argument("--no-auto-resume", action="store_true",
                    help="do not pick up <save-folder>/weights/last.pth after an elastic / launcher restart")
    return ap.parse_args(argv)


def restart_count():
    """How often this job has been restarted by its launcher (torchrun's elastic agent exports
    TORCHELASTIC_RESTART_COUNT, the self-launcher DTP_RESTART_COUNT)."""
    return max(int(os.environ.get("TORCHELASTIC_RESTART_COUNT", "0") or 0),
               int(os.environ.get("DTP_RESTART_COUNT", "0") or 0))


def pick_snapshot(args):
    """Explicit --snapshot wins; otherwise a RESTARTED job resumes from the last snapshot it wrote
    (the decision is taken inside each restarted worker, not once by the
    shell before the first launch)."""
    if args.snapshot or args.no_auto_resume:
        return args.snapshot
    if restart_count() == 0:
        return None
    wdir = os.path.join(args.save_folder, "weights")
    last = os.path.join(wdir, "last.pth")
    if os.path.isfile(last):
        return last
    # runs without validation write checkpoint_epoch_N.pth instead of last.pth (reference
    # trainer/trainer.py:163-167): resume from the newest of those
    import re
    best_n, best_path = -1, None
    for name in (os.listdir(wdir) if os.path.isdir(wdir) else []):
        m = re.fullmatch(r"checkpoint_epoch_(\d+)\.pth", name)
        if m and int(m.group(1)) > best_n:
            best_n, best_path = int(m.group(1)), os.path.join(wdir, name)
    if best_path is None:
        print("WARNING: restarted job (restart %d) found no snapshot under %s - training starts from epoch 0"
              % (restart_count(), wdir), flush=True)
    return best_path


if __name__ == "__main__":
    args = parse_args()
    if args.nproc > 0 and "WORLD_SIZE" not in os.environ:
        # self-launch (reference run.sh:9-14 uses torchrun): this parent never touches the GPU
        import sys
        from distributed_training_pytorch_amd.launch import spawn_workers
        sys.exit(spawn_workers(os.path.abspath(__file__), sys.argv[1:], args.nproc, max_restarts=args.max_restarts))
    logger = Logger("VGG16", file="%s/logfile.log" % args.save_folder)

    backend = args.backend or ("nccl" if torch.cuda.is_available() else "gloo")
    ExampleTrainer.ddp_setup(backend=backend)

    train_path = "synthetic:%d" % args.synthetic_size if args.synthetic else args.train_path
    val_path = "synthetic:%d" % max(args.synthetic_size // 4, 1) if args.synthetic else args.val_path
    trainer = ExampleTrainer(train_path=train_path,
                             val_path=val_path,
                             labels=args.labels.split(","),
                             height=args.image_size,
                             width=args.image_size,
                             max_epoch=args.epochs,
                             batch_size=args.batch_size,
                             pin_memory=True,
                             have_validate=not args.no_validate,
                             save_best_for=("accuracy", "geq"),
                             save_period=args.save_period,
                             save_folder=args.save_folder,
                             snapshot_path=pick_snapshot(args),
                             logger=logger,
                             lr=args.lr,
                             batch_norm=args.batch_norm, fp8_classifier=args.fp8_classifier, uint8_input=args.uint8_input,
                             grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else torch.float32,
                             grad_accum_steps=args.grad_accum,
                             num_workers=args.num_workers,
                             max_steps_per_epoch=args.steps,
                             progress=not args.no_progress, use_hip_graph=args.hip_graph, cache_dataset_on_device=args.device_cache,
                             shard_optimizer=args.shard_optimizer, ring_loader=args.ring_loader,
                             bucket_optimizer=not args.no_bucket_optimizer, seed=args.seed,
                             metrics_file="%s/metrics.jsonl" % args.save_folder)

    trainer.train()

    ExampleTrainer.destroy_process()
