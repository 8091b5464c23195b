"""Identity of the kernel sources a profile was made from: sha256 over csrc/ (sources, headers, Makefile) and include/nqe.h.
profiles/rNN/pmc_traffic_*.json record it; bench.py reports `roofline.traffic` only when it matches the running tree."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_rev() -> str:
    h = hashlib.sha256()
    d = os.path.join(ROOT, "naive_query_engine_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".hpp")) or f == "Makefile")
    files.append(os.path.join(ROOT, "include", "nqe.h"))
    for p in files:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_rev())
