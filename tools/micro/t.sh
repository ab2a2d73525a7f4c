for g in 256 512 768 1024; do echo "== WGs $g"; SBA_BS_WGS=$g timeout -k 10 100 python tools/probe_time.py 2>&1 | grep "backsub"; done
