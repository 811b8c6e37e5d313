for wl in c3 c3 sso; do python tools/ab_inproc.py $wl 1000000 2>&1 | tail -7; done
