cd $GRAFT_REPO_ROOT
for e in "DTP_NONE=1" "DTP_PREFETCH_DERIVED=0"; do
  echo "== $e"
  env $e timeout -k 10 400 python bench/determinism_probe.py --trials 200 --steps 3 2>&1 | grep -E "probe:|first differing|starts in backward" | cut -c1-60 | sort | uniq -c | sort -rn | head -6
done
