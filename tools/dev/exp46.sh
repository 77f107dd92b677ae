timeout -k 10 600 python tools/dev/ab.py build_exp/final1.so build_exp/final_bs4.so
