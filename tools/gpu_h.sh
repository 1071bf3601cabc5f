for th in 23 25 27 29 31; do
  echo "== cfg5 VRP_TILE_MIN_SEL=$th"; VRP_TILE_MIN_SEL=$th python tools/step_probe.py 1,100,2048,0,1 2>/dev/null | grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' '; echo
done
for th in 27 29 31 33; do
  echo "== TSP 8192x40 VRP_TILE_MIN_SEL=$th"; VRP_TILE_MIN_SEL=$th python tools/step_probe.py 0,40,8192 2>/dev/null | grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' '; echo
done
for th in 22 24 26 28; do
  echo "== VRP 8192x40 VRP_TILE_MIN_SEL=$th"; VRP_TILE_MIN_SEL=$th python tools/step_probe.py 1,40,8192 2>/dev/null | grep -o '"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' '; echo
done
