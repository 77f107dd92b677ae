python tools/dev/ab.py build_exp/stC.so build_exp/noatom.so build_exp/rmw.so
