"""Device lists for the multi-device tests: on a box with several GPUs the tests spread their
engines over DISTINCT devices (real peer copies, a real second RCCL rank), on a one-GPU box they are what they were.
Pure Python (no GPU call here): tests/test_distributed_cpu.py checks the construction with a faked device count."""


def device_lists(n_dev: int, n_have: int):
    """The lists a test runs for `n_dev` listed devices on a box with `n_have` GPUs: always device 0 listed n_dev
    times (every code path but a real peer link), and - when the box has more than one GPU - the round-robin list
    [0, 1, ..., n_have - 1, 0, ...] cut to n_dev entries, so that neighbouring shards live on different devices."""
    assert n_dev >= 1 and n_have >= 1
    lists = [[0] * n_dev]
    spread = [i % n_have for i in range(n_dev)]
    if spread != lists[0]:
        lists.append(spread)
    return lists


def rccl_world(n_have: int, cap: int = 8) -> int:
    """Ranks the RCCL test starts: one per GPU, at most `cap` (RCCL refuses two ranks on one device)."""
    return max(1, min(n_have, cap))
