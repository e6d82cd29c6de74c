#!/usr/bin/env python3
"""One line per launch of scripts/startup_probe.sh: python scripts/startup_probe_summary.py <name> [seconds the invocation took]"""
import json, re, sys
n = sys.argv[1]; took = sys.argv[2] if len(sys.argv) > 2 else "?"
p = "gpurun_out/startup/%s" % n
try:
    b = json.loads([l for l in open(p + "/bench.json").read().splitlines() if l.strip()][-1]); e = b["e2e"]; d = e["dropin"]
    err = open(p + "/bwa-meme_dropin.stderr").read()
    st = re.search(r"index staged in HBM in ([0-9.]+) s", err)
    wt = re.search(r"worker buffers \+ fwd/rc text ([0-9.]+) s, HBM index ([0-9.]+) s", err)
    rg = re.search(r"Reading IO time \(Reference Genome\) avg: ([0-9.]+)", err)
    print("%-16s invocation %s s | wall %.2f  staged %s  waited-for-index %s  genome-read %s  process %s  identical %s  reference cached %s" % (
        n, took, d["wall_s"], st.group(1) if st else "?", wt.group(2) if wt else "?", rg.group(1) if rg else "?", d["process_s"], e["sam_identical"],
        "cached" in e["reference"]), flush=True)
except Exception as ex:
    print(n, "FAILED after", took, "s:", repr(ex), flush=True)
    try: print(open(p + "/bench.stderr").read()[-500:])
    except OSError: pass
