echo "--- no-records"; timeout -k 10 300 python tools/dev/ab.py --no-records build_exp/coop_sf.so
echo "--- 49152 games"; timeout -k 10 300 python tools/dev/ab.py --num-envs=49152 build_exp/coop_sf.so
