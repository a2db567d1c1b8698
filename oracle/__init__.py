"""CPU ORACLE -- test infrastructure, not product code (see oracle/hj_oracle.h).
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() import this."""
