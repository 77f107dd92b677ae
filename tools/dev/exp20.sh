python tools/dev/ab.py build_exp/stA.so build_exp/stB1.so build_exp/stB3.so build_exp/stB13.so
