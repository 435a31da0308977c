import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
same = np.array_equal(a["sum"].view(np.uint32), b["sum"].view(np.uint32))
print("sum identical:", same, "| rays", int(a["rays"]), int(b["rays"]), "| hist identical:", np.array_equal(a["hist"], b["hist"]))
sys.exit(0 if same and a["rays"] == b["rays"] else 1)
