"""Best-effort log of the errors the GPU tests observe (never fails a test).  One JSON file PER TEST NAME PREFIX under
gpurun_out/parity_observed/ - so a partial run (pytest -k ...) rewrites only the entries it produced and a copy of the directory
into profiles/ never loses another run's record (a 3-entry partial run had replaced the ~60-entry file)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "gpurun_out", "parity_observed")


def observed(name, value):
    try:
        os.makedirs(DIR, exist_ok=True)
        group = re.split(r"[\[\]:/ ]", name)[0].split("_vs_")[0][:48] or "misc"
        p = os.path.join(DIR, group + ".json")
        rec = json.load(open(p)) if os.path.exists(p) else {}
        rec[name] = value
        with open(p, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass
