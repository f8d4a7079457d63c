#!/bin/bash
# session baseline: full GPU suite + default bench line on HEAD
exec < /dev/null
O=gpurun_out/r5base; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
( timeout 2400 python -m pytest tests -q -m gpu -x -W ignore 2>&1 | tail -15 ) > $O/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
tail -c 600 $O/bench.json; echo; tail -5 $O/suite.log; tail -1 $O/smoke.log
