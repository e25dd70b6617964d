#!/bin/bash
# env_ab.sh <tag> VAR=val ...: same-box A/B of runtime environment settings on the default bench (3 alternating rounds)
tag=$1; shift
out=gpurun_out/r05_env_$tag; mkdir -p $out
for rep in 1 2 3; do
  echo "== none rep $rep"; timeout 300 python3 bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  for e in "$@"; do
    echo "== $e rep $rep"; env $e timeout 300 python3 bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-loader-path --global-batch 0 --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  done
done
