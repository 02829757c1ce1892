# Trainer path fed by the pinned ring (fresh host-generated uint8 batches every step) and config 5 through the Trainer
set -e
O=gpurun_out/r3trainer
mkdir -p $O
rm -rf /tmp/runs_r3
timeout -k 10 400 python main.py --synthetic --synthetic-size 16384 --epochs 4 --batch-size 1024 --no-validate --grad-dtype bf16 --ring-loader --lr 0.001 --save-folder /tmp/runs_r3 --no-progress > $O/trainer.log 2>&1
cp /tmp/runs_r3/metrics.jsonl $O/85_trainer_ring_loader_metrics_round3.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r3trainer/85_trainer_ring_loader_metrics_round3.jsonl"):
    d = json.loads(l)
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k in ("epoch", "train_img_per_s", "img_per_s", "throughput", "data_wait_ms_per_step", "data_starved_ms_per_step", "step_ms", "train_loss")})
PY
tail -3 $O/trainer.log
