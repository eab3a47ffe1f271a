#!/usr/bin/env python3
"""End to end on the UNCHANGED script's surface: the 9-mer high-precision stages exactly as
scripts/umgap-analyse.sh:257-264 + :299-304 chain them -- a resident `umgap prot2kmer2lca -m -o -s <socket>` server, and per
sample  <proteins> | nc -NU socket | umgap seedextend -g1 -s3 | umgap uniq -d / | umgap taxa2agg -l2 -a 'lca*' taxons  --
text in, text out, four processes, beside
  * the same four stages of the CPU oracle (oracle/_build/umgap_oracle, one thread per stage) on a bounded sample, and
  * the fused `umgap prot2kmer2agg` (one process, same output bytes).
`nc` is not in the image: tools/native/unixcat.c plays its part (stdin into the socket, the answer into the pipe; a Python loop
when there is no C compiler).  run() is imported by bench.py (extras "end_to_end" / "cpu_baseline_end_to_end"); as a script it prints one JSON line.
    python tools/end_to_end.py [n_pairs_gpu] [n_pairs_cpu]"""
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
UMGAP = os.path.join(ROOT, "umgap_amd", "bin", "umgap")
ORACLE = os.path.join(ROOT, "oracle", "_build", "umgap_oracle")


def _make_inputs(work, n_pairs, n_frag, flen=50):
    """Synthetic proteome (n_frag fragments of flen residues, index = all their 9-mer windows), taxonomy, and protein FASTA
    of n_pairs read pairs (2 mutated proteome fragments each): what FGSpp hands to the socket."""
    import ctypes as C

    import torch

    from umgap_amd import _lib, synth
    from umgap_amd.api import Index, _dev_ptr, _stream_ptr
    L = _lib.load()
    dev = torch.device("cuda", 0)
    ids, par, valid, rank = synth.synthetic_taxonomy(100_000, 140_000)
    tsv = os.path.join(work, "taxons.tsv")
    synth.write_taxonomy_tsv(tsv, ids, par, valid, rank)
    offs = torch.arange(0, (n_frag + 1) * flen, flen, dtype=torch.int64, device=dev)
    prot = torch.empty(n_frag * flen, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_peptides_dev(0, n_frag, 7, _dev_ptr(offs), _dev_ptr(prot), 0, _stream_ptr(None)))
    taxa = torch.from_numpy(ids.view("int32")).to(dev)[torch.arange(n_frag, device=dev) % len(ids)].contiguous()
    h = C.c_void_p()
    _lib.check(L.umgap_index_build_windows9_dev(_dev_ptr(prot), _dev_ptr(offs), _dev_ptr(taxa), n_frag, flen, n_frag * (flen - 8), 0.5, 0, C.byref(h)))
    idx = os.path.join(work, "ninemer.umgapix")
    ix = Index(h.value)
    n_index_keys = ix.num_keys
    ix.save(idx)
    ix.close()
    reads = torch.empty(n_pairs * 2 * flen, dtype=torch.uint8, device=dev)
    _lib.check(L.umgap_synth_prot_reads_dev(_dev_ptr(prot), n_frag, flen, 44, 0, n_pairs, 2, 30, _dev_ptr(reads), 0, _stream_ptr(None)))
    rows = reads.cpu().numpy().reshape(2 * n_pairs, flen)
    fasta = os.path.join(work, "proteins.fa")
    with open(fasta, "wb") as f:
        for r in range(2 * n_pairs):
            f.write(b">read%d/%d\n" % (r // 2, r % 2 + 1))
            f.write(rows[r].tobytes())
            f.write(b"\n")
    del reads, prot, offs, taxa
    torch.cuda.empty_cache()
    return idx, tsv, fasta, n_index_keys


def _head(path, n_records, out):
    """First n_records FASTA records (2 lines each here) of `path` into `out`."""
    with open(path, "rb") as fi, open(out, "wb") as fo:
        for _ in range(2 * n_records):
            fo.write(fi.readline())
    return out


def _socket_pipeline(sock_path, fasta, tsv, out_path, exe=UMGAP):
    """<fasta> | nc -NU sock | seedextend -g1 -s3 | uniq -d / | taxa2agg -l2 -a lca* tsv > out ; returns seconds."""
    t0 = time.perf_counter()
    with open(out_path, "wb") as fo:
        p3 = subprocess.Popen([exe, "taxa2agg", "-l", "2", "-a", "lca*", tsv], stdin=subprocess.PIPE, stdout=fo)
        p2 = subprocess.Popen([exe, "uniq", "-d", "/"], stdin=subprocess.PIPE, stdout=p3.stdin)
        p1 = subprocess.Popen([exe, "seedextend", "-g", "1", "-s", "3"], stdin=subprocess.PIPE, stdout=p2.stdin)
        p3.stdin.close()
        p2.stdin.close()
        nc = _unixcat()
        if nc:                           # `nc -NU socket`: stdin -> socket (sending side shut at EOF), socket -> stdout
            with open(fasta, "rb") as fi:
                p0 = subprocess.Popen([nc, sock_path], stdin=fi, stdout=p1.stdin)
            p1.stdin.close()
            rcs = [p.wait() for p in (p0, p1, p2, p3)]
            assert rcs == [0, 0, 0, 0], rcs
            return time.perf_counter() - t0
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)     # no C compiler: a Python loop plays nc (slower than the chain)
        s.connect(sock_path)

        def feed():
            with open(fasta, "rb") as f:
                while True:
                    b = f.read(1 << 22)
                    if not b:
                        break
                    s.sendall(b)
            s.shutdown(socket.SHUT_WR)
        th = threading.Thread(target=feed)
        th.start()
        while True:
            b = s.recv(1 << 22)
            if not b:
                break
            p1.stdin.write(b)
        p1.stdin.close()
        th.join()
        s.close()
        rcs = [p.wait() for p in (p1, p2, p3)]
    assert rcs == [0, 0, 0], rcs
    return time.perf_counter() - t0


def _unixcat():
    """tools/native/unixcat (built on first use): the part `nc -NU` plays in the script.  None without a C compiler."""
    import shutil
    exe = os.path.join(ROOT, "tools", "native", "unixcat")
    src = exe + ".c"
    if os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src):
        return exe
    cc = shutil.which("cc") or shutil.which("gcc")
    if not cc:
        return None
    try:
        subprocess.run([cc, "-O2", "-pthread", "-o", exe, src], check=True)
    except (OSError, subprocess.CalledProcessError):
        return None
    return exe


def _usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _plain_pipeline(exe, index, fasta, tsv, out_path, lookup_threads=0):
    t0 = time.perf_counter()
    cmd = "%s prot2kmer2lca -o %s%s < %s | %s seedextend -g 1 -s 3 | %s uniq -d / | %s taxa2agg -l 2 -a 'lca*' %s > %s" % (
        exe, "-j %d " % lookup_threads if lookup_threads else "", index, fasta, exe, exe, exe, tsv, out_path)
    subprocess.run(cmd, shell=True, check=True)
    return time.perf_counter() - t0


def run(n_pairs_gpu=1_000_000, n_pairs_cpu=40_000, n_frag=250_000, work="/tmp/umgap_e2e"):
    os.makedirs(work, exist_ok=True)
    idx, tsv, fasta, n_index_keys = _make_inputs(work, n_pairs_gpu, n_frag)
    res = {"surface": "protein FASTA in (2 x 50 residues per pair, %.0f MB per %d pairs) -> one taxon per pair out; index of %d 9-mers" % (
        os.path.getsize(fasta) / 1e6, n_pairs_gpu, n_index_keys)}
    # ---- the resident server of the script, then one sample through the socket ----
    sock = os.path.join(work, "socket")
    if os.path.exists(sock):
        os.unlink(sock)
    t0 = time.perf_counter()
    server = subprocess.Popen([UMGAP, "prot2kmer2lca", "-m", "-o", "-s", sock, idx], stdout=subprocess.DEVNULL)
    try:
        while not os.path.exists(sock):
            if server.poll() is not None:
                raise RuntimeError("the prot2kmer2lca server exited with %d" % server.returncode)
            time.sleep(0.02)
        res["server_start_s"] = round(time.perf_counter() - t0, 2)
        out = os.path.join(work, "out.socket")
        _socket_pipeline(sock, _head(fasta, 2000, os.path.join(work, "warm.fa")), tsv, out + ".warm")   # first connection: page in
        dt = _socket_pipeline(sock, fasta, tsv, out)
        res["four_process_socket"] = {"pairs": n_pairs_gpu, "s": round(dt, 3), "pairs_per_s": round(n_pairs_gpu / dt, 1),
                                      "chain": "umgap prot2kmer2lca -m -o -s (resident) | seedextend -g1 -s3 | uniq -d / | taxa2agg -l2 -a lca* (scripts/umgap-analyse.sh:257-264,299-304)"}
    finally:
        server.terminate()
        server.wait()
    # ---- the fused command on the same input ----
    fused = os.path.join(work, "out.fused")
    env = dict(os.environ, UMGAP_TIMING="1")
    for _ in range(2):
        t0 = time.perf_counter()
        with open(fasta, "rb") as fi, open(fused, "wb") as fo:
            p = subprocess.run([UMGAP, "prot2kmer2agg", "-g", "1", "-s", "3", "-d", "/", "-l", "2", "-a", "lca*", idx, tsv], stdin=fi, stdout=fo, stderr=subprocess.PIPE,
                               env=env, check=True)
        dt = time.perf_counter() - t0
    res["fused_command"] = {"pairs": n_pairs_gpu, "s": round(dt, 3), "pairs_per_s": round(n_pairs_gpu / dt, 1), "start_up_included": True,
                            "same_output_as_four_process": open(fused, "rb").read() == open(out, "rb").read()}
    # the command's own clock (UMGAP_TIMING): from "index and taxonomy are resident" to "the last read is written" = its steady state
    import re
    m = re.search(r"stream ([0-9.]+) s", p.stderr.decode(errors="replace"))
    if m and float(m.group(1)) > 0:
        stream_s = float(m.group(1))
        res["fused_command"].update({"stream_s": stream_s, "start_up_s": round(dt - stream_s, 3), "steady_state_pairs_per_s": round(n_pairs_gpu / stream_s, 1)})
    # ---- the CPU twin: the oracle's four stages through the same text pipes, bounded sample; its index is the same table
    # as a TSV (what `umgap buildindex` eats).  Start-up (TSV parse) is measured on an empty sample and subtracted. ----
    idx_tsv = os.path.join(work, "ninemer.tsv")
    with open(idx_tsv, "wb") as fo:
        subprocess.run([UMGAP, "printindex", idx], stdout=fo, check=True)
    sample = _head(fasta, 2 * n_pairs_cpu, os.path.join(work, "sample.fa"))
    empty = os.path.join(work, "empty.fa")
    open(empty, "wb").close()
    cores = _usable_cores()
    t_empty = _plain_pipeline(ORACLE, idx_tsv, empty, tsv, os.path.join(work, "out.oracle0"), cores)
    t_cpu = _plain_pipeline(ORACLE, idx_tsv, sample, tsv, os.path.join(work, "out.oracle"), cores)
    prod_sample = os.path.join(work, "out.sample")
    _plain_pipeline(UMGAP, idx, sample, tsv, prod_sample)
    marginal = max(1e-6, t_cpu - t_empty)
    cpu = {"value": round(n_pairs_cpu / marginal, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
           "sample": "%d pairs through the oracle's prot2kmer2lca -o | seedextend -g1 -s3 | uniq -d / | taxa2agg -l2 -a lca* as four piped processes "
                     "(the lookup stage on %d threads, as the reference's rayon par_bridge; the three filters one thread each, as the reference's): %.2f s, "
                     "of which %.2f s start-up (index TSV parse) measured on an empty input and subtracted" % (n_pairs_cpu, cores, t_cpu, t_empty),
           "same_output_as_product": open(prod_sample, "rb").read() == open(os.path.join(work, "out.oracle"), "rb").read()}
    res["speedup_four_process_vs_cpu_twin"] = round(res["four_process_socket"]["pairs_per_s"] / cpu["value"], 1)
    return res, cpu


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:3]]
    r, c = run(*a)
    print(json.dumps({"end_to_end": r, "cpu_baseline_end_to_end": c}))
