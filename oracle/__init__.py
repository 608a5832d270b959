"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/lba_oracle.c header).

May be imported from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() only; never from orb_slam3_study_kr_amd/.
"""
