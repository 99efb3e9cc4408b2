#!/bin/bash
# The round's remaining GPU minutes: two more seeds of the parity fuzz tool and a determinism soak on the final library.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/lastfuzz; mkdir -p $O
timeout 150 python tools/parity_fuzz.py 115 2718 > $O/parity_fuzz_tool_seed2718.txt 2>&1
timeout 150 python tools/parity_fuzz.py 115 16180 > $O/parity_fuzz_tool_seed16180.txt 2>&1
timeout 80 python tools/soak.py 45 > $O/soak_final.txt 2>&1; echo "soak exit $?" >> $O/soak_final.txt
tail -1 $O/parity_fuzz_tool_seed2718.txt; tail -1 $O/parity_fuzz_tool_seed16180.txt; tail -3 $O/soak_final.txt
