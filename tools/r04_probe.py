import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GSC_MAX_BATCH", "256"); os.environ.setdefault("GSC_WINDOW_Z", "6"); os.environ.setdefault("GSC_W_TABLE_GB", "8"); os.environ.setdefault("GSC_FEW_Z_GB", "0")
os.environ["GSC_TRACE_HOST"] = "1"
import gsc_loader
from bench import golden
g = gsc_loader.load()
assert g.init_algorithm(0, golden("pk.chacha20"), golden("r1cs.chacha20"))
print(g.describe(0))
