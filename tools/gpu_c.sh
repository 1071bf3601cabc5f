for w in 1 2 4; do echo "waves=$w"; VRP_PERSISTENT_WAVES=$w python tools/step_probe.py 0,20,512 1,40,1024 2>/dev/null | grep workload | cut -c1-250; done
echo default; python tools/step_probe.py 1,40,1024 1,40,768 2>/dev/null | grep workload | cut -c1-250
