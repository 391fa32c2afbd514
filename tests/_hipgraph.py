"""Test helper: the node types of a captured hipGraph, through the HIP runtime torch already loaded (ctypes on the
libamdhip64 of /proc/self/maps).  Round 4 found that a memset NODE (a captured hipMemsetAsync) is not ordered before the
kernel node after it when the graph is replayed: the library enqueues kernels only, and the tests
assert that of every capture the package makes."""
import contextlib
import ctypes

import torch

KERNEL, MEMCPY, MEMSET, HOST, GRAPH, EMPTY, WAIT_EVENT, EVENT_RECORD = range(8)   # hipGraphNodeType
NAMES = {KERNEL: "kernel", MEMCPY: "memcpy", MEMSET: "memset", HOST: "host", GRAPH: "child graph", EMPTY: "empty",
         WAIT_EVENT: "wait event", EVENT_RECORD: "event record"}


def _hip():
    for line in open("/proc/self/maps"):
        if "libamdhip64.so" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 is not mapped into this process (is torch a ROCm build?)")


@contextlib.contextmanager
def keep_graphs():
    """Every torch.cuda.CUDAGraph() made inside keeps its hipGraph_t (raw_cuda_graph) and is collected in the yielded list."""
    orig, made = torch.cuda.CUDAGraph, []

    def factory(*a, **kw):
        g = orig(keep_graph=True)
        made.append(g)
        return g
    torch.cuda.CUDAGraph = factory
    try:
        yield made
    finally:
        torch.cuda.CUDAGraph = orig


def node_types(cg):
    """{type name: count} of the nodes of a torch.cuda.CUDAGraph made under keep_graphs()."""
    hip = _hip()
    graph = ctypes.c_void_p(cg.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    assert hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) == 0
    out = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) == 0
        k = NAMES.get(t.value, f"type {t.value}")
        out[k] = out.get(k, 0) + 1
    return out
