"""TEST INFRASTRUCTURE: ctypes bindings for the CPU oracle (limg_oracle.c) and the real reference build (_ref/).

Only tests/, tools/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package."""
