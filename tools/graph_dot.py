"""Topology of a captured training step (DOT file written with FSNET_AMD_GRAPH_DOT=<path>): node list, longest chain,
width, and for a named kernel the nearest ancestors — to check which branches the capture really left independent.
    python tools/graph_dot.py step.dot [kernel-substring ...]"""
import collections
import re
import sys


def load(path):
    txt = open(path).read()
    nodes = {}
    for m in re.finditer(r'"graph_0_node_(\d+)"\[[^\]]*?label="\{\s*(\w+)\s*\|(.*?)\}"\];', txt, re.S):
        nid, kind, body = int(m.group(1)), m.group(2), m.group(3)
        name = kind
        km = re.search(r"\{ID \| \d+ \| (\S+?)\\<", body)
        if km:
            name = km.group(1).replace("_ZN12_GLOBAL__N_1", "").lstrip("0123456789")[:40]
        nodes[nid] = name
    edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_0_node_(\d+)"\s*->\s*"graph_0_node_(\d+)"', txt)]
    return nodes, edges


def main():
    nodes, edges = load(sys.argv[1])
    succ, pred = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in edges:
        succ[a].append(b); pred[b].append(a)
    print("%d nodes, %d edges, roots %s" % (len(nodes), len(edges), [n for n in nodes if not pred[n]][:10]))
    # topological levels
    level = {}
    order = sorted(nodes)
    indeg = {n: len(pred[n]) for n in nodes}
    q = collections.deque(n for n in nodes if indeg[n] == 0)
    topo = []
    while q:
        n = q.popleft(); topo.append(n)
        level[n] = 1 + max((level[p] for p in pred[n]), default=-1)
        for s in succ[n]:
            indeg[s] -= 1
            if indeg[s] == 0:
                q.append(s)
    width = collections.Counter(level.values())
    print("depth (longest chain) %d; widest level %d; nodes with >1 successor: %d, >1 predecessor: %d"
          % (max(level.values()) + 1, max(width.values()), sum(len(v) > 1 for v in succ.values()),
             sum(len(v) > 1 for v in pred.values())))
    for pat in sys.argv[2:]:
        for n in order:
            if pat in nodes[n]:
                anc = set(); st = list(pred[n])
                while st:
                    x = st.pop()
                    if x not in anc:
                        anc.add(x); st.extend(pred[x])
                hist = collections.Counter(nodes[a][:28] for a in anc)
                print("node %d %s: level %d, %d ancestors, direct preds %s" % (n, nodes[n], level[n], len(anc),
                      [(p, nodes[p][:24]) for p in pred[n]]))
                print("    ancestor kernels:", dict(hist.most_common(12)))
    if len(sys.argv) == 2:
        for n in order:
            print(n, level[n], nodes[n], "<-", pred[n])


main()
