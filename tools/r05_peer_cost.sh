#!/bin/bash
# where the kernel-side cost of seven peers goes (one GPU, loopback rings): default build / peer stores at agent scope /
# no peer stores at all — experiment slots of tools/build_alt.sh 1614 "-DTDS_PEER_SCOPE=__HIP_MEMORY_SCOPE_AGENT" "-DTDS_X_PEER_NOSTORE"
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r05c}
mkdir -p $O
for alt in ${ALTS:-0 1 2}; do
  OPT=""; [ $alt != 0 ] && OPT="--option alt_build=$alt"
  timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline $OPT > $O/bench_alt$alt.json 2> $O/bench_alt$alt.err
  python3 - $O/bench_alt$alt.json $alt <<'P'
import json,sys
d=json.load(open(sys.argv[1]))
print("alt",sys.argv[2],"value %.4g"%d["value"], " ".join("%s %.4g (%.3f)"%(k[:24],d[k]["value"],d[k].get("ratio_to_value",0)) for k in ("one_rank_with_exchange","one_rank_with_exchange_7_loopback_peers","steady_state_1000")))
P
done
