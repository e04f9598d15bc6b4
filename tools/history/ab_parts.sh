#!/bin/bash
# inference bench line against the number of concurrent sub-batches: usage  bash tools/ab_parts.sh [model] [batch] [img]
cd $GRAFT_REPO_ROOT
for p in 1 2 4 8 1 2 4 8; do
  r=$(python bench.py --mode infer --model ${1:-lemevit_base} --batch ${2:-128} --img ${3:-224} --infer-parts $p --no-cpu-baseline --no-kernel-timing --no-issue-probe --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "parts $p: $r ms"
done
