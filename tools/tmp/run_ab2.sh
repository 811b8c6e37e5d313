for wl in c3 sso c5 c5site; do python tools/ab_inproc.py $wl 1000000 2>&1 | tail -6; done
