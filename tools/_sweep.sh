for o in "streams=2,blocks=1" "streams=2" "streams=3,blocks=1" "streams=4,blocks=1" "blocks=1" "min_batch=8" "min_batch=2" "static=7" "static=5"; do
  echo "== $o"; timeout -k 10 120 python bench.py --steps 3 --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_frame'])" || exit 1
done
