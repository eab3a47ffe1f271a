#!/usr/bin/env python3
"""End-to-end throughput of the `umgap` command on the GPU box: text in -> text out, the stages of
the 9-mer high-precision pipeline (scripts/umgap-analyse.sh:299-304 minus FGSpp) as separate
processes connected by pipes, exactly as the script runs them.  Prints one JSON line."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from umgap_amd import synth

UMGAP = os.path.join(ROOT, "umgap_amd", "bin", "umgap")
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/umgap_cli"
os.makedirs(work, exist_ok=True)
n_tax, n_keys = 100_000, 50_000_000
w = synth.SyntheticWorkload(n_tax=n_tax, n_keys=n_keys)
idx, tsv = os.path.join(work, "ninemer.umgapix"), os.path.join(work, "taxons.tsv")
w.index.save(idx)
synth.write_taxonomy_tsv(tsv, w.ids, w.parents, w.valid, w.rank)
keys = w.reads(0, n_pairs).cpu().numpy().view(np.uint64)
# pept2lca surface: one record per read end (42 9-mers, one per line)
letters = np.zeros((len(keys), 10), np.uint8)
for j in range(9):
    letters[:, j] = ((keys >> np.uint64(5 * (8 - j))) & np.uint64(31)).astype(np.uint8) + 65
letters[:, 9] = 10
body = letters.reshape(2 * n_pairs, 42 * 10)
fasta = os.path.join(work, "in.fa")
with open(fasta, "wb") as f:
    for r in range(2 * n_pairs):
        f.write(b">read%d/%d\n" % (r // 2, r % 2 + 1))
        f.write(body[r].tobytes())
size = os.path.getsize(fasta)


def timed(cmd, stdin_path, stdout_path):
    t0 = time.perf_counter()
    with open(stdin_path, "rb") as fi, open(stdout_path, "wb") as fo:
        subprocess.run(cmd, stdin=fi, stdout=fo, check=True)
    return time.perf_counter() - t0


res = {"pairs": n_pairs, "input_MB": round(size / 1e6, 1)}
h, se, un, out = (os.path.join(work, x) for x in ("hits.fa", "se.fa", "un.fa", "out.fa"))
res["pept2lca_s"] = round(timed([UMGAP, "pept2lca", "-o", idx], fasta, h), 3)
res["seedextend_s"] = round(timed([UMGAP, "seedextend", "-g", "1", "-s", "3"], h, se), 3)
res["uniq_s"] = round(timed([UMGAP, "uniq", "-d", "/"], se, un), 3)
res["taxa2agg_s"] = round(timed([UMGAP, "taxa2agg", "-l", "2", "-a", "lca*", tsv], un, out), 3)
t0 = time.perf_counter()
pipe = "%s pept2lca -o %s < %s | %s seedextend -g1 -s3 | %s uniq -d / | %s taxa2agg -l2 -a 'lca*' %s > %s.pipe" % (UMGAP, idx, fasta, UMGAP, UMGAP, UMGAP, tsv, out)
subprocess.run(pipe, shell=True, check=True)
res["pipeline_s"] = round(time.perf_counter() - t0, 3)
res["pipeline_pairs_per_s"] = round(n_pairs / res["pipeline_s"], 1)
res["same_output"] = open(out, "rb").read() == open(out + ".pipe", "rb").read()
res["records_out"] = open(out, "rb").read().count(b">")

# ---- surface B: protein fragments in (what FGSpp hands to prot2kmer2lca), 2 x 50 residues per pair ----------
import ctypes as C

from umgap_amd import _lib
from umgap_amd.api import Index, _dev_ptr, _stream_ptr

L = _lib.load()
dev = torch.device("cuda", 0)
n_frag, flen = 1_000_000, 50
offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device=dev)
prot = torch.empty(n_frag * flen, dtype=torch.uint8, device=dev)
_lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
taxa = w.d_taxon_of_row[torch.arange(n_frag, device=dev) % w.n_tax].contiguous()
hnd = C.c_void_p()
_lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * (flen - 8), 0.5, 0, C.byref(hnd)))
idx_b = os.path.join(work, "proteome.umgapix")
Index(hnd.value).save(idx_b)
reads = torch.empty(n_pairs * 2 * flen, dtype=torch.uint8, device=dev)
_lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_pairs, 2, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
rows = reads.cpu().numpy().reshape(2 * n_pairs, flen)
fasta_b = os.path.join(work, "in_prot.fa")
with open(fasta_b, "wb") as f:
    for r in range(2 * n_pairs):
        f.write(b">read%d/%d\n" % (r // 2, r % 2 + 1))
        f.write(rows[r].tobytes())
        f.write(b"\n")
del w, reads, prot, offs, taxa
torch.cuda.empty_cache()
resb = {"input_MB": round(os.path.getsize(fasta_b) / 1e6, 1)}
t0 = time.perf_counter()
pipe = "%s prot2kmer2lca -o %s < %s | %s seedextend -g1 -s3 | %s uniq -d / | %s taxa2agg -l2 -a 'lca*' %s > %s.b_pipe" % (UMGAP, idx_b, fasta_b, UMGAP, UMGAP, UMGAP, tsv, out)
subprocess.run(pipe, shell=True, check=True)
resb["four_stage_pipe_s"] = round(time.perf_counter() - t0, 3)
resb["four_stage_pairs_per_s"] = round(n_pairs / resb["four_stage_pipe_s"], 1)
for rep in range(2):   # second run: page cache warm, like the pipe above after the file was just written
    resb["fused_s"] = round(timed([UMGAP, "prot2kmer2agg", "-g", "1", "-s", "3", "-d", "/", "-l", "2", "-a", "lca*", idx_b, tsv], fasta_b, out + ".b_fused"), 3)
resb["fused_pairs_per_s"] = round(n_pairs / resb["fused_s"], 1)
# 5 x the input: start-up (HIP init, index upload, taxonomy) amortised; phase times from UMGAP_TIMING
fasta_b5 = os.path.join(work, "in_prot5.fa")
with open(fasta_b5, "wb") as fo:
    blob = open(fasta_b, "rb").read()
    for _ in range(5):
        fo.write(blob)
del blob
t0 = time.perf_counter()
with open(fasta_b5, "rb") as fi, open(out + ".b_fused5", "wb") as fo:
    pr = subprocess.run([UMGAP, "prot2kmer2agg", "-g", "1", "-s", "3", "-d", "/", "-l", "2", "-a", "lca*", idx_b, tsv], stdin=fi, stdout=fo, stderr=subprocess.PIPE,
                        env=dict(os.environ, UMGAP_TIMING="1"), check=True)
resb["fused_5x_s"] = round(time.perf_counter() - t0, 3)
resb["fused_5x_pairs_per_s"] = round(5 * n_pairs / resb["fused_5x_s"], 1)
resb["fused_marginal_pairs_per_s"] = round(4 * n_pairs / max(1e-9, resb["fused_5x_s"] - resb["fused_s"]), 1)
resb["fused_5x_phases"] = pr.stderr.decode().strip().splitlines()[-1]
resb["same_output"] = open(out + ".b_pipe", "rb").read() == open(out + ".b_fused", "rb").read()
fused_out = open(out + ".b_fused", "rb").read()
resb["records_out"] = fused_out.count(b">")
resb["classified_below_root"] = resb["records_out"] - fused_out.count(b"\n1\n")
res["surface_b"] = resb
print(json.dumps(res))
