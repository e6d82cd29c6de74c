#!/bin/bash
# Round 6, GPU call E.  SURVEY 8(a) B3-B7 (the ring of band columns in k_bsw_lane: parity, both ways) and 8(f)1 (the extension stage in rounds on repeat-dense reads: probe).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python scripts/r06_ext_rounds_probe.py 64 100000 > $O/ext_rounds_probe.log 2>&1; echo "probe rc $?" >> $O/ext_rounds_probe.log
tail -40 $O/ext_rounds_probe.log | cut -c1-250
timeout 1500 python -m pytest tests/test_gpu_bsw.py tests/test_gpu_ext.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log | cut -c1-300
for circ in 1 0; do for rl in 250 400; do MEME_TUNING="bsw_circ=$circ" python scripts/bsw_probe.py 1 $rl 1 2>&1 | tail -1 | sed "s/^/circ=$circ /"; done; done > $O/bsw_probe.log 2>&1; cat $O/bsw_probe.log | cut -c1-250
