for o in "" "event_threshold=32" "event_threshold=24" "new_threshold=4" "new_threshold=8" "log_cap=96" "log_cap=48" "march_stop=16" "event_threshold=32 new_threshold=4" ""; do
  timeout -k 5 100 python scripts/bench_ne.py xos1 291 1000000 - $o 2>&1 | head -1 | cut -c40-175
done
