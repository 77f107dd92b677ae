timeout -k 10 600 python tools/dev/ab.py build_exp/coop.so build_exp/coop_nost.so build_exp/coop_nox.so build_exp/coop_nost_nox.so
