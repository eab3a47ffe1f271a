#!/bin/bash
# Where the four-process chain loses its time: phase times of its first and last stage inside the chain (UMGAP_TIMING), and the
# middle stages alone with a file or a pipe on either side.   bash tools/chain_timing.sh [n_pairs]
set -e
N=${1:-8000000}
python - $N <<'PY'
import os, sys
sys.path.insert(0, 'tools')
import end_to_end as E
os.makedirs('/tmp/umgap_e2e', exist_ok=True)
idx, tsv, fasta, _ = E._make_inputs('/tmp/umgap_e2e', int(sys.argv[1]), 250000)
open('/tmp/umgap_e2e/paths', 'w').write("%s\n%s\n%s\n" % (idx, tsv, fasta))
PY
IDX=$(sed -n 1p /tmp/umgap_e2e/paths); TSV=$(sed -n 2p /tmp/umgap_e2e/paths); FA=$(sed -n 3p /tmp/umgap_e2e/paths)
U=umgap_amd/bin/umgap
W=/tmp/umgap_e2e
t() { local s=$(date +%s%N); "$@"; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms"; }
export UMGAP_TIMING=1
chain() { $U prot2kmer2lca -o $IDX < $FA 2> $W/t1.err | $U seedextend -g 1 -s 3 2> $W/t2.err | $U uniq -d / 2> $W/t3.err | $U taxa2agg -l 2 -a 'lca*' $TSV 2> $W/t4.err > $W/out.chain; }
echo "chain (prot2kmer2lca opens the index itself), $N pairs: $(t chain) $(t chain)"
for i in 1 4; do grep -h "stage A" $W/t$i.err | tail -1; done
unset UMGAP_TIMING
$U prot2kmer2lca -o $IDX < $FA > $W/s1.out
se() { $U seedextend -g 1 -s 3; }
uq() { $U uniq -d /; }
a() { se < $W/s1.out > $W/s2.out; }
b() { cat $W/s1.out | se > /dev/null; }
c() { se < $W/s1.out | cat > /dev/null; }
d() { cat $W/s1.out | se | cat > /dev/null; }
echo "seedextend  file->file $(t a) | pipe->null $(t b) | file->pipe $(t c) | pipe->pipe $(t d)   ($(du -m $W/s1.out | cut -f1) MB in)"
a2() { uq < $W/s2.out > $W/s3.out; }
b2() { cat $W/s2.out | uq > /dev/null; }
d2() { cat $W/s2.out | uq | cat > /dev/null; }
echo "uniq        file->file $(t a2) | pipe->null $(t b2) | pipe->pipe $(t d2)   ($(du -m $W/s2.out | cut -f1) MB in)"
e() { cat $W/s1.out | se | uq | cat > /dev/null; }
echo "cat | seedextend | uniq | cat: $(t e)"
f() { cat $W/s1.out | cat > /dev/null; }
echo "cat | cat: $(t f)"
