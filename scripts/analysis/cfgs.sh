for cfg in "xos1 1 10000000 -" "ellip_l9 1 4000000 5.0" "ellip_l9 1 4000000 -" "cone 1 1000000 -" "xos1 4 2000000 -" "xos1 12 1000000 -" "xos1 291 1000000 -"; do
  timeout -k 10 120 python scripts/bench_ne.py $cfg $EXTRA 2>&1 | head -1 | sed "s/: [0-9]* exit slots, [0-9]* started,//; s/ (wall.*//"
done
