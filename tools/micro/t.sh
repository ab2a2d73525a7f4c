for f in 0 1 2 3 4 6; do echo "== flags $f"; SBA_DBGF=$f SBA_SCHUR_DEBUG=1 timeout -k 10 100 python tools/probe_time.py 2>&1 | grep -A3 "schur stamps\|schur  " | grep "it  1\|schur  "; done
