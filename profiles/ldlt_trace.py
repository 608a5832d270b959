import sys
sys.path.insert(0, "/root/repo")
from orb_slam3_study_kr_amd import lba, synth
sv = lba.LbaSolver(0)
w = synth.make_config2(100)
w.max_iterations = 2
sv.upload([w]); sv.optimize()
sv.close()
