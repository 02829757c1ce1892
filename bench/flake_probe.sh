cd $GRAFT_REPO_ROOT
for a in "--bn" "--fp8" "--adamw --grad-bf16" "--dropout 0.3" "--graph"; do
  echo "== $a"
  timeout -k 10 300 python bench/determinism_probe.py --trials 60 --steps 3 $a 2>&1 | grep -E "probe:|first differing|starts in backward|Error|error" | cut -c1-160 | sort | uniq -c | sort -rn | head -5
done
