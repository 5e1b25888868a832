"""Host-side placement of a rank: one process per GPU, its threads on ONE block of cores (a CCD: eight cores sharing an L3) of the
GPU's NUMA node, a small intra-op thread pool.

Why: the eager Detection iteration (Detection/train_aug_sat_muti_advt.py:70-172 — ~3 800 launches, 16 host reads, host-side
`torch.randperm` sampling between them) is issued by two threads (Python's and the autograd engine's) plus the HIP runtime's; on a
2 x 64-core host with 256 hardware threads and PyTorch's default 128-thread intra-op pool they migrate freely: 53 - 65 ms per iteration
for ONE binary, process to process.  On eight cores of one CCD with four intra-op threads: 47.6 - 48.6 ms, every run
(profiles/r06_host_placement.txt).  The reference's own launcher (`torch.distributed.run`) sets OMP_NUM_THREADS=1 per rank for the same
reason; it does not pin.  Nothing here changes arithmetic.

`place_rank()` is called by the command-line entry points (main_perturb.py, main_learnable.py) and by bench.py BEFORE the GPU is
initialised (threads the HIP runtime starts later inherit the mask).  AFAN_HOST_PIN=0 switches it off, AFAN_HOST_CPUS="64-71" names the
block, AFAN_HOST_THREADS the intra-op threads."""
import glob
import os


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _gpu_numa_nodes():
    """NUMA node of every AMD GPU this process can open, in PCI address order (the HIP runtime's default device order): the render nodes
    under /dev/dri that are accessible (a container is usually handed a subset of the machine's GPUs through them); if none can be
    told apart that way, every AMD display / accelerator PCI function."""
    seen = []
    for rd in glob.glob("/sys/class/drm/renderD*"):
        try:
            dev = os.path.realpath(rd + "/device")
            if open(dev + "/vendor").read().strip() != "0x1002" or not os.access("/dev/dri/" + os.path.basename(rd), os.R_OK | os.W_OK):
                continue
            seen.append((os.path.basename(dev), int(open(dev + "/numa_node").read().strip())))
        except (OSError, ValueError):
            continue
    if seen:
        return [n for _, n in sorted(seen)]
    nodes = []
    for dev in sorted(glob.glob("/sys/bus/pci/devices/*")):
        try:
            if open(dev + "/vendor").read().strip() != "0x1002" or not open(dev + "/class").read().strip().startswith(("0x03", "0x12")):
                continue
            nodes.append(int(open(dev + "/numa_node").read().strip()))
        except (OSError, ValueError):
            continue
    return nodes


def rank_cpus(local_rank=0, cores=8):
    """The block of `cores` physical cores for this rank: on its GPU's NUMA node (node 0 if the topology cannot be read), block number
    `local_rank` of that node's list (wrapping), restricted to what the process may run on.  None: nothing sensible to choose."""
    allowed = sorted(os.sched_getaffinity(0))
    env = os.environ.get("AFAN_HOST_CPUS")
    if env:
        want = [c for c in _parse_cpulist(env) if c in allowed]
        return want or None
    nodes = _gpu_numa_nodes()
    node = nodes[local_rank] if local_rank < len(nodes) and nodes[local_rank] >= 0 else (nodes[0] if nodes and nodes[0] >= 0 else 0)
    try:
        node_cpus = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except OSError:
        node_cpus = allowed
    # physical cores first: a node's list is "first threads of its cores, then their SMT siblings" — keep the first half when SMT is on
    try:
        smt = open("/sys/devices/system/cpu/smt/active").read().strip() == "1"
    except OSError:
        smt = False
    phys = node_cpus[:len(node_cpus) // 2] if smt and len(node_cpus) >= 2 * cores else node_cpus
    phys = [c for c in phys if c in allowed]
    if len(phys) < cores:
        return None
    blocks = len(phys) // cores
    b = local_rank % blocks
    return phys[b * cores:(b + 1) * cores]


def place_rank(local_rank=None, cores=8, threads=None):
    """Pin this process to its block and size torch's intra-op pool; returns a dict describing what was done (bench.py prints it) with the
    previous affinity / thread count under "restore" for `restore()`."""
    import torch
    info = {"pinned": False}
    if os.environ.get("AFAN_HOST_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return info
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    before = sorted(os.sched_getaffinity(0))
    cpus = rank_cpus(local_rank, cores)
    if not cpus:
        return info
    nthreads = int(os.environ.get("AFAN_HOST_THREADS", threads if threads is not None else min(4, len(cpus))))
    info["restore"] = (before, torch.get_num_threads())
    os.sched_setaffinity(0, cpus)
    torch.set_num_threads(max(1, nthreads))
    info.update(pinned=True, cpus=f"{cpus[0]}-{cpus[-1]}" if cpus == list(range(cpus[0], cpus[-1] + 1)) else ",".join(map(str, cpus)),
                intra_op_threads=max(1, nthreads))
    return info


def restore(info):
    """Undo place_rank() for host-side work that wants the whole machine (bench.py's CPU baseline)."""
    import torch
    if info and info.get("restore"):
        cpus, n = info["restore"]
        os.sched_setaffinity(0, cpus)
        torch.set_num_threads(n)
