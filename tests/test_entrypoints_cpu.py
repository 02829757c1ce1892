"""CPU tier for the user-facing flow: image-folder dataset, a single-process
ExampleTrainer run (validation, best tracking, snapshots, LR schedule, metrics file, resume) and
eval.py on the snapshot it wrote — the reference's main.py -> trainer -> eval.py path."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _make_image_folder(root, labels, per_class, size=(40, 36)):
    from PIL import Image
    rng = np.random.RandomState(0)
    for li, lb in enumerate(labels):
        os.makedirs(os.path.join(root, lb), exist_ok=True)
        for i in range(per_class):
            a = rng.randint(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)
            a[..., li % 3] = 255 - 40 * li                      # a learnable per-class colour cast
            Image.fromarray(a).save(os.path.join(root, lb, "img_%02d.png" % i))


def test_image_folder_dataset(tmp_path):
    from distributed_training_pytorch_amd.data.image_folder import ExampleDataset
    labels = ["cat", "dog", "snake"]
    _make_image_folder(str(tmp_path), labels, 4)
    val = ExampleDataset(str(tmp_path), labels, 24, 32, phase="val")
    assert len(val) == 12
    x, y = val[0]
    assert x.shape == (3, 24, 32) and x.dtype == torch.float32 and 0 <= y < 3
    assert sorted(set(val[i][1] for i in range(12))) == [0, 1, 2]
    # deterministic, rank-independent order (quirk #3): a second instance sees the same samples
    val2 = ExampleDataset(str(tmp_path), labels, 24, 32, phase="val")
    assert all(torch.equal(val[i][0], val2[i][0]) and val[i][1] == val2[i][1] for i in range(12))
    # ImageNet normalisation (reference dataset/example_dataset.py:45)
    raw = ExampleDataset(str(tmp_path), labels, 24, 32, phase="val", normalize=False)[0][0]
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    assert torch.allclose(x, (raw - mean) / std, atol=1e-6)
    # uint8 mode: raw pixels for the fused device-side normalisation
    u = ExampleDataset(str(tmp_path), labels, 24, 32, phase="val", uint8=True)[0][0]
    assert u.dtype == torch.uint8 and torch.allclose(u.float() / 255.0, raw, atol=1e-6)
    # train phase: augmentations keep shape/dtype and are reproducible for a given seed
    t1 = ExampleDataset(str(tmp_path), labels, 24, 32, phase="train", seed=5)
    t2 = ExampleDataset(str(tmp_path), labels, 24, 32, phase="train", seed=5)
    a, b = [t1[i][0] for i in range(6)], [t2[i][0] for i in range(6)]
    assert all(p.shape == (3, 24, 32) and torch.isfinite(p).all() for p in a)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    # ... a function of (seed, epoch, index) only: independent of call order (any rank / worker
    # serving sample i draws the same stream; one shared RNG gave every rank identical streams and
    # made the result depend on who asked first), different between samples and between epochs
    t3 = ExampleDataset(str(tmp_path), labels, 24, 32, phase="train", seed=5)
    assert torch.equal(t3[4][0], a[4]) and torch.equal(t3[1][0], a[1])         # reverse order, same pixels
    t3.set_epoch(1)
    assert any(not torch.equal(t3[i][0], a[i]) for i in range(6))
    t4 = ExampleDataset(str(tmp_path), labels, 24, 32, phase="train", seed=6)
    assert any(not torch.equal(t4[i][0], a[i]) for i in range(6))
    with pytest.raises(Exception):
        ExampleDataset(str(tmp_path / "missing"), labels, 24, 32, phase="val")[0]


def test_trainer_single_process_flow_and_eval(tmp_path, capsys):
    from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer
    from distributed_training_pytorch_amd.utils.logger import Logger
    save = str(tmp_path / "runs")
    metrics = str(tmp_path / "metrics.jsonl")
    kw = dict(train_path="synthetic:24", val_path="synthetic:12", labels=["a", "b", "c"], height=32, width=32,
              batch_size=8, pin_memory=True, have_validate=True, save_best_for=("accuracy", "geq"),
              save_period=1, save_folder=save, progress=False, lr=0.01, milestones=(1, 5), device="cpu",
              metrics_file=metrics)
    logger = Logger("T", file=os.path.join(save, "logfile.log"))
    tr = ExampleTrainer(max_epoch=2, snapshot_path=None, logger=logger, **kw)
    assert tr.world_size == 1 and tr.local_batch_size == 8
    tr.train()
    files = sorted(os.listdir(os.path.join(save, "weights")))
    assert files == ["best.pth", "last.pth"], files
    from distributed_training_pytorch_amd.utils.checkpoint import load_snapshot
    snap = load_snapshot(os.path.join(save, "weights", "last.pth"))      # the safe (weights_only) loader
    assert sorted(k for k in snap if k in ("epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict")) \
        == ["epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict"]
    assert snap["epoch"] == 2 and len(snap["model_state_dict"]) == 32
    assert abs(tr.scheduler.get_last_lr()[0] - 0.001) < 1e-12
    allrecs = [json.loads(l) for l in open(metrics)]
    recs = [r for r in allrecs if "images_per_sec" in r]                # per-epoch training records
    assert len(recs) == 2 and all(r["images_per_sec"] > 0 and "train_ce_loss" in r for r in recs)
    assert sum(1 for r in allrecs if "val_accuracy" in r) == 2          # one validation record per epoch
    assert all(k in recs[0] for k in ("data_wait_ms_per_step", "host_step_ms_per_step", "allreduce_exposed_ms_per_step"))
    log = open(os.path.join(save, "logfile.log")).read()
    for needle in ("[GPU0] Epoch 2/2", "VALIDATE RESULTS", "The BEST model is at EPOCH", "Saved model at epoch 2!",
                   "THE NEXT LEARNING RATE VALUE IS", "THROUGHPUT", "Finished!"):
        assert needle in log, needle
    # resume continues where the snapshot stopped, with the persisted best fitness (quirks #4, #7)
    tr2 = ExampleTrainer(max_epoch=3, snapshot_path=os.path.join(save, "weights", "last.pth"), logger=logger, **kw)
    assert tr2.cur_epoch == 2 and tr2.best_fitness["epoch"] == tr.best_fitness["epoch"]
    assert torch.equal(tr2.raw_model.param_arena.flat_master, tr.raw_model.param_arena.flat_master)
    tr2.train()
    assert "Epoch 3/3" in open(os.path.join(save, "logfile.log")).read()
    logger.close()
    # config validation (quirks #5, #11)
    with pytest.raises(ValueError):
        ExampleTrainer(max_epoch=1, **dict(kw, save_period=0))
    with pytest.raises(ValueError):
        ExampleTrainer(max_epoch=1, **dict(kw, save_best_for=None))

    # eval.py consumes model_state_dict of the snapshot (reference eval.py:47-51, 69-72)
    import eval as eval_script
    acc1, acck, n = eval_script.main(["--synthetic", "12", "--model", os.path.join(save, "weights", "last.pth"),
                                      "--image-size", "32", "--device", "cpu", "--labels", "a,b,c", "--batch-size", "4"])
    assert n == 12 and 0.0 <= acc1 <= acck <= 1.0
    assert "EVALUATION ACCURACY RESULTS: TOP-1=" in capsys.readouterr().out


def test_grad_accumulation_flushes_the_trailing_short_window(tmp_path):
    """3 steps per epoch with grad_accum_steps=2: the third micro-batch is a window of its own —
    synchronised, stepped with its loss scaled by 1/1 (not 1/2) — instead of being silently
    dropped at the epoch boundary; equal to the hand-rolled schedule on the same batches."""
    from distributed_training_pytorch_amd.models.vgg16 import VGG16
    from distributed_training_pytorch_amd.trainer.example_trainer import ExampleTrainer

    class NoDropTrainer(ExampleTrainer):
        def build_model(self):
            return VGG16(3, len(self.labels), True, dropout=0.0)

    kw = dict(train_path="synthetic:12", val_path="synthetic:4", labels=["a", "b", "c"], height=32, width=32,
              batch_size=4, pin_memory=False, have_validate=False, save_period=1, progress=False, lr=0.01,
              device="cpu", max_epoch=1)
    torch.manual_seed(11)
    tr = NoDropTrainer(save_folder=str(tmp_path / "a"), grad_accum_steps=2, **kw)
    assert [tr.accum_window(i) for i in range(3)] == [2, 2, 2]          # before an epoch: no length known
    tr.train()
    assert tr.steps_this_epoch == 3
    assert [tr.should_sync(i) for i in range(3)] == [False, True, True]
    assert [tr.accum_window(i) for i in range(3)] == [2, 2, 1]
    assert tr.optimizer._steps == 2                                      # window [0, 1] and the flushed window [2]

    torch.manual_seed(11)
    ref = NoDropTrainer(save_folder=str(tmp_path / "b"), grad_accum_steps=1, **kw)
    ref.train_dataloader.sampler.set_epoch(0)
    batches = list(ref.train_dataloader)
    assert len(batches) == 3
    model, opt, crit = ref.model, ref.optimizer, ref.criterion
    model.train()
    opt.zero_grad()
    (crit(model(batches[0][0]), batches[0][1]) / 2).backward()
    (crit(model(batches[1][0]), batches[1][1]) / 2).backward()
    opt.step()
    opt.zero_grad()
    crit(model(batches[2][0]), batches[2][1]).backward()
    opt.step()
    a, b = tr.raw_model.param_arena.flat_master, ref.raw_model.param_arena.flat_master
    assert torch.allclose(a, b, atol=1e-7), float((a - b).abs().max())
