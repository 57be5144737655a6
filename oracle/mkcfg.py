#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: write a machine config (INI) for the reference's stock x86 pthread/HC
target -- PolicyDomain=HC, Scheduler=COMMON + heuristic HC, root scheduler-object WST,
CommPlatform=None (section/keys as read by ocr/src/driver/ocr-driver.c:220-287 and
ocr/src/machine-description/ocr-machine-description.c:593-1140).  The reference's own generator
(ocr/scripts/Configs/config-generator.py) is Python 2 and is not run."""
import sys

def cfg(workers: int, guid: str = "COUNTED_MAP", db: str = "Regular", alloc_mb: int = 64, heuristic: str = "HC", workpile: str = "HC") -> str:
    """heuristic / workpile = "GPU" and guid = "GPUHASH" name the factories of integration/ (oracle/Makefile.mi355x builds)"""
    w = workers - 1
    rng = f"0-{w}"
    size = alloc_mb << 20
    L = []
    def sec(section, **kv):
        L.append(f"[{section}]\n" + "".join(f"\t{k}\t=\t{v}\n" for k, v in kv.items()))
    sec("General", version="1.1.0")
    sec("GuidType0", name=guid); sec("GuidInst0", id=0, type=guid)
    sec("PolicyDomainType0", name="HC")
    sec("PolicyDomainInst0", id=0, type="HC", worker=rng, scheduler=0, allocator=0, commapi=0, guid=0, parent=0,
        location=0, taskfactory="HC", tasktemplatefactory="HC", datablockfactory=db, eventfactory="HC")
    sec("TaskType0", name="HC"); sec("TaskTemplateType0", name="HC"); sec("DataBlockType0", name=db); sec("EventType0", name="HC")
    sec("MemPlatformType0", name="malloc"); sec("MemPlatformInst0", id=0, type="malloc", size=int(size * 1.05))
    sec("MemTargetType0", name="shared"); sec("MemTargetInst0", id=0, type="shared", size=int(size * 1.05), memplatform=0)
    sec("AllocatorType0", name="mallocproxy"); sec("AllocatorInst0", id=0, type="mallocproxy", size=size, memtarget=0)
    sec("CommPlatformType0", name="None"); sec("CommPlatformInst0", id=0, type="None")
    sec("CommApiType0", name="Handleless"); sec("CommApiInst0", id=0, type="Handleless", commplatform=0)
    sec("CompPlatformType0", name="pthread", stacksize=0); sec("CompPlatformInst0", id=rng, type="pthread", stacksize=0)
    sec("CompTargetType0", name="PASSTHROUGH"); sec("CompTargetInst0", id=rng, type="PASSTHROUGH", compplatform=rng)
    sec("WorkerType0", name="HC"); sec("WorkerInst0", id=0, type="HC", workertype="master", comptarget=0)
    if workers > 1:
        sec("WorkerInst1", id=f"1-{w}", type="HC", workertype="slave", comptarget=f"1-{w}")
    sec("WorkPileType0", name=workpile); sec("WorkPileInst0", id=rng, type=workpile, dequetype="WORK_STEALING_DEQUE")
    names = ["NULL", "WST", "DEQ", "LIST", "MAP", "PDSPACE", "DBSPACE", "DBTIME", "PR_WSH", "BIN_HEAP"]
    for i, n in enumerate(names):
        if n == "WST": sec(f"SchedulerObjectType{i}", name=n, kind="root")
        else: sec(f"SchedulerObjectType{i}", name=n)
    sec("SchedulerObjectInst0", id=0, type="WST")
    for i, n in enumerate(["NULL", "HC", "ST", "PRIORITY", "STATIC"] + ([heuristic] if heuristic == "GPU" else [])):
        sec(f"SchedulerHeuristicType{i}", name=n)
    sec("SchedulerHeuristicInst0", id=0, type=heuristic)
    sec("SchedulerType0", name="COMMON")
    sec("SchedulerInst0", id=0, type="COMMON", workpile=rng, workeridfirst=0, schedulerObject=0,
        config="comp0:plc0:comm0", schedulerHeuristic=0)
    return "\n".join(L)

if __name__ == "__main__":
    workers = int(sys.argv[1]); guid = sys.argv[2] if len(sys.argv) > 2 else "COUNTED_MAP"
    gpu = guid == "GPUHASH"
    sys.stdout.write(cfg(workers, guid, heuristic="GPU" if gpu else "HC", workpile="GPU" if gpu else "HC"))
