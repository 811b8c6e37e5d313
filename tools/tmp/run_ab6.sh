for wl in c3both ssoboth c5both; do python tools/ab_inproc.py $wl 1000000 2>&1 | tail -8; done
