"""Constants shared by bench.py and its two helper modules (tools/bench_extra.py, tools/bench_multirank.py)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
N_SIMDS = 1024         # 256 CUs x 4 SIMDs
SEED = 20251031
DT = 1.0 / 252.0
RB = dict(S0=100.0, r=0.04, xi=0.04, H=0.1, eta=1.9, rho=-0.9)
