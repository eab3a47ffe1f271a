import json, os, subprocess, sys, time
sys.path.insert(0, "tools")
import end_to_end as E
n = 8_000_000
work = "/tmp/umgap_e2e"; os.makedirs(work, exist_ok=True)
idx, tsv, fasta, _ = E._make_inputs(work, n, 250_000)
res = {}
for mode in ("copy", "splice"):
    os.environ["UNIXCAT_SPLICE"] = "1" if mode == "splice" else "0"
    sock = os.path.join(work, "sock_" + mode)
    if os.path.exists(sock): os.unlink(sock)
    server = subprocess.Popen([E.UMGAP, "prot2kmer2lca", "-m", "-o", "-s", sock, idx], stdout=subprocess.DEVNULL)
    try:
        while not os.path.exists(sock): time.sleep(0.02)
        out = os.path.join(work, "out." + mode)
        E._socket_pipeline(sock, E._head(fasta, 2000, os.path.join(work, "warm.fa")), tsv, out + ".warm")
        best = min(E._socket_pipeline(sock, fasta, tsv, out) for _ in range(2))
        res["socket_relay_" + mode] = round(n / best / 1e6, 3)
    finally:
        server.terminate(); server.wait()
t = min(E._plain_pipeline(E.UMGAP, idx, fasta, tsv, os.path.join(work, "out.plain")) for _ in range(2))
res["plain_pipe_no_socket_start_up_included"] = round(n / t / 1e6, 3)
res["same"] = open(os.path.join(work, "out.plain"), "rb").read() == open(os.path.join(work, "out.copy"), "rb").read() == open(os.path.join(work, "out.splice"), "rb").read()
print(json.dumps(res))
