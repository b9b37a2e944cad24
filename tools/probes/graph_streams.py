"""Read a DEBUG_HIP_GRAPH_DOT_PRINT=1 dump of a captured step (the HIP runtime writes graph_<n>_dot_print_<k> into the
working directory at instantiation) and print, per executor stream, the runs of nodes in launch order — which chain of the
step the graph executor put on which of its streams.

    cd somewhere && DEBUG_HIP_GRAPH_DOT_PRINT=1 python /root/repo/bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-kernel-profile
    python tools/probes/graph_streams.py somewhere/graph_*_dot_print_*          [--nodes: every node with its edges]
"""
import re
import sys


def short(name):
    m = re.search(r"N_1\d+([a-z0-9_]+?)(_kernel|E|I)", name)
    return (m.group(1) if m else name)[:30]


def main(path, every):
    s = open(path).read()
    nodes = re.findall(r'"graph_\d+_node_(\d+)"\[[^\]]*label="\d+\n([^\n]*)\n(?:\([^\n]*\n)?StreamId:(\d+)\nSignalIsRequired: (\w+)', s)
    succ = {}
    for a, b in re.findall(r'"graph_\d+_node_(\d+)" -> "graph_\d+_node_(\d+)"', s):
        succ.setdefault(int(a), []).append(int(b))
    print("%s: %d nodes, %d edges, %d signals" % (path, len(nodes), sum(map(len, succ.values())),
                                                  sum(1 for n in nodes if n[3] == "true")))
    if every:
        for n, name, sid, sig in nodes:
            print("%4d  stream %s %s %-30s -> %s" % (int(n), sid, "S" if sig == "true" else " ", short(name), succ.get(int(n), [])))
        return
    runs = []
    for n, name, sid, sig in nodes:
        if runs and runs[-1][0] == sid:
            runs[-1][2] = int(n)
            runs[-1][4] = short(name)
        else:
            runs.append([sid, int(n), int(n), short(name), short(name)])
    for sid, a, b, first, last in runs:
        print("  stream %s  nodes %3d-%3d (%3d)  %s .. %s" % (sid, a, b, b - a + 1, first, last))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for p in args:
        main(p, "--nodes" in sys.argv)
