timeout -k 10 600 python tools/dev/ab.py build_exp/coop_sf.so build_exp/sf_cond.so build_exp/sf_cond_nopeel.so
