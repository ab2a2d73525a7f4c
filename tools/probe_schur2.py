import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.probe_schur as ps
ps.run(20, 200, 0.7)
ps.run(20, 64, 1.0)
