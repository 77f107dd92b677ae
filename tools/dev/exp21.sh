python tools/dev/ab.py build_exp/stA.so build_exp/stC.so
