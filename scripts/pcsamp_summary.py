#!/usr/bin/env python3
"""Aggregate rocprofv3 PC-sampling CSVs: samples per source line and per instruction of k_seed."""
import sys, os, csv, collections, glob, re
root = sys.argv[1]
files = [f for f in glob.glob(os.path.join(root, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
print("files:", files)
csv.field_size_limit(1 << 30)
for f in files:
    with open(f, newline="") as fh:
        rd = csv.reader(fh)
        hdr = next(rd)
        print("==", f, "columns:", hdr)
        idx = {h: i for i, h in enumerate(hdr)}
        ci = idx.get("Instruction"); cc = idx.get("Instruction_Comment")
        extra = [h for h in hdr if h not in ("Sample_Timestamp", "Exec_Mask", "Dispatch_Id", "Instruction", "Instruction_Comment", "Correlation_Id")]
        by_line = collections.Counter(); by_ins = collections.Counter(); n = 0
        lanes = collections.Counter(); by_extra = {h: collections.Counter() for h in extra}
        first = []
        for row in rd:
            if len(first) < 3: first.append(row)
            ins = row[ci] if ci is not None else ""
            com = row[cc] if cc is not None else ""
            n += 1
            m = re.search(r"([\w\.]+):(\d+)", com)
            key = (m.group(1), int(m.group(2))) if m else ("?", 0)
            by_line[key] += 1
            by_ins[(key, ins)] += 1
            if "Exec_Mask" in idx:
                try: lanes[bin(int(row[idx["Exec_Mask"]])).count("1") // 8] += 1
                except Exception: pass
            for h in extra:
                by_extra[h][row[idx[h]]] += 1
        print("samples", n); print("first rows", first)
        print("active lanes (x8 buckets):", sorted(lanes.items()))
        for h in extra:
            if len(by_extra[h]) <= 40: print("column", h, by_extra[h].most_common(40))
        src = {}
        def line(k):
            fn, ln = k
            if fn not in src:
                cand = glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bwa-meme_amd", "csrc", fn))
                src[fn] = open(cand[0]).read().split("\n") if cand else []
            return src[fn][ln - 1].strip()[:100] if 0 < ln <= len(src[fn]) else ""
        print("-- by source line")
        for k, c in by_line.most_common(70): print("%6.2f%% %s:%d | %s" % (100.0 * c / n, k[0], k[1], line(k)))
        print("-- by instruction")
        for (k, ins), c in by_ins.most_common(60): print("%6.2f%% %s:%d  %s" % (100.0 * c / n, k[0], k[1], ins))
