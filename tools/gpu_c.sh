for v in 0 1; do
  if [ $v = 1 ]; then export VRP_TILE_V1=1; else unset VRP_TILE_V1; fi
  python tools/tile_phase_probe.py 1 100 2048 3 1 2>&1 | tail -1
  python tools/tile_phase_probe.py 0 40 8192 3 0 2>&1 | tail -1
done
