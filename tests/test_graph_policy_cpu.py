"""Admission / eviction policy of the hipGraph cache (selfmask_amd/graphs.py), checked without a GPU: one-off keys
never capture, recurring keys capture at the third sighting, eviction is LRU and resets the evicted key's count
(FIFO eviction + a surviving sighting count made native-resolution runs capture on almost every call)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "salient-object-detection_amd"))

from selfmask_amd.graphs import GraphCachePolicy  # noqa: E402


def drive(policy, key):
    what = policy.decide(key)
    if what == "capture":
        policy.admit(key, f"graph:{key}")
    return what


def test_admission_after_three_sightings():
    p = GraphCachePolicy(max_entries=4, admit_after=2)
    assert [drive(p, "a") for _ in range(5)] == ["eager", "eager", "capture", "replay", "replay"]
    assert len(p) == 1


def test_one_off_shapes_never_capture():
    p = GraphCachePolicy(max_entries=4, admit_after=2)
    assert all(drive(p, ("shape", i)) == "eager" for i in range(200))
    assert len(p) == 0


def test_lru_eviction_and_reset_count():
    evicted = []
    p = GraphCachePolicy(max_entries=2, admit_after=1, on_evict=lambda k, e: evicted.append(k))
    for k in ("a", "b"):
        assert [drive(p, k) for _ in range(2)] == ["eager", "capture"]
    assert drive(p, "a") == "replay"          # a becomes the youngest
    assert [drive(p, "c") for _ in range(2)] == ["eager", "capture"]
    assert evicted == ["b"] and len(p) == 2   # least recently used went, not the oldest capture
    assert drive(p, "a") == "replay"
    # the evicted key must earn its capture again (its sighting count was dropped with it)
    assert [drive(p, "b") for _ in range(2)] == ["eager", "capture"]
    assert evicted == ["b", "c"]


def test_many_keys_do_not_thrash():
    # 30 image sizes x 3 streams cycling through 8 slots: captures stay rare compared with calls
    p = GraphCachePolicy(max_entries=8, admit_after=2)
    calls = captures = 0
    for rep in range(10):
        for size in range(30):
            for stream in range(3):
                calls += 1
                captures += drive(p, (size, stream)) == "capture"
    assert captures <= calls // 3 and p.evictions <= captures


def test_seen_table_is_bounded_and_clear_evicts():
    evicted = []
    p = GraphCachePolicy(max_entries=2, admit_after=0, max_seen=16, on_evict=lambda k, e: evicted.append(k))
    assert drive(p, "x") == "capture"
    for i in range(100):
        p.decide(("never-again", i)) if False else None
    q = GraphCachePolicy(max_entries=2, admit_after=5, max_seen=16)
    for i in range(100):
        q.decide(i)
    assert len(q._seen) <= 16
    p.clear()
    assert evicted == ["x"] and len(p) == 0 and drive(p, "x") == "capture"
