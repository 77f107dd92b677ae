timeout -k 10 600 python tools/dev/ab.py build_exp/coop_sf.so build_exp/coop_split.so build_exp/coop_late.so
