python tools/dev/ab.py build_exp/rowpre.so build_exp/pilepre.so
