"""Which fork/join topologies does hipGraph stream capture accept on this ROCm/PyTorch?  (diagnostic)"""
import subprocess, sys, textwrap
CASES = {
 "nested_fork_reuse_before_join": """
    with torch.cuda.stream(sC):
        for i in range(6):
            s = side[i % 2]; s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s): b[i].add_(1)
            a.add_(1)
        for s in side: torch.cuda.current_stream().wait_stream(s)
 """,
 "nested_fork_join_each": """
    with torch.cuda.stream(sC):
        for i in range(6):
            s = side[i % 2]; s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s): b[i].add_(1)
            a.add_(1)
            torch.cuda.current_stream().wait_stream(s)
 """,
 "nested_fork_join_before_reuse": """
    with torch.cuda.stream(sC):
        used = set()
        for i in range(6):
            s = side[i % 2]
            if s in used: torch.cuda.current_stream().wait_stream(s)
            s.wait_stream(torch.cuda.current_stream()); used.add(s)
            with torch.cuda.stream(s): b[i].add_(1)
            a.add_(1)
        for s in side: torch.cuda.current_stream().wait_stream(s)
 """,
 "prefork_from_main_then_wait_on_sC": """
    for s in side: s.wait_stream(main)
    with torch.cuda.stream(sC):
        for i in range(6):
            s = side[i % 2]; s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s): b[i].add_(1)
            a.add_(1)
    for s in side: main.wait_stream(s)
 """,
 "prefork_join_into_sC": """
    for s in side: s.wait_stream(main)
    with torch.cuda.stream(sC):
        for i in range(6):
            s = side[i % 2]; s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s): b[i].add_(1)
            a.add_(1)
        for s in side: torch.cuda.current_stream().wait_stream(s)
 """,
 "flat_fork_reuse_before_join": """
    sC.wait_stream(main)
    for i in range(6):
        s = side[i % 2]; s.wait_stream(main)
        with torch.cuda.stream(s): b[i].add_(1)
        a.add_(1)
    for s in side: main.wait_stream(s)
 """,
}
TEMPLATE = """
import torch
a = torch.zeros(1024, device='cuda'); b = [torch.zeros(1024, device='cuda') for _ in range(6)]
sC = torch.cuda.Stream(); side = [torch.cuda.Stream(), torch.cuda.Stream()]
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    a.add_(1)
    sC.wait_stream(main)
{body}
    main.wait_stream(sC)
g.replay(); g.replay(); torch.cuda.synchronize()
print('OK', float(a[0]), [float(x[0]) for x in b])
"""
for name, body in CASES.items():
    code = TEMPLATE.format(body=textwrap.indent(textwrap.dedent(body), "    "))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    err = r.stderr.strip().replace("amdgpu.ids", "ids")[-160:]
    out = r.stdout.strip()[-80:]
    print(name, "->", out if r.returncode == 0 else "FAILED rc=%d %s" % (r.returncode, err), flush=True)
