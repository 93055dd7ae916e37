"""Host-side helpers of the engines that need no GPU."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "retinanet-tensorflow2.x_amd"))


def _graph(depths):
    g = types.SimpleNamespace()
    g.convs = {f"c{i}": {"k": k, "cin": cin} for i, (k, cin) in enumerate(depths)}
    ops = [{"op": "conv", "conv": f"c{i}", "group": "grp"} for i in range(len(depths))]
    return g, ops


def test_split_by_depth_groups_the_deep_segments(monkeypatch):
    """engine.split_by_depth: a grouped forward launch whose segments differ 2x or more in K depth becomes two launches, the
    deep segments first; homogeneous groups and single ops stay one launch; RNET_GROUP_SPLIT=0 switches it off."""
    from retinanet.model.engine import split_by_depth
    monkeypatch.delenv("RNET_GROUP_SPLIT", raising=False)
    g, ops = _graph([(1, 2048), (1, 512), (1, 1024), (1, 2048)])       # the FPN lateral convs (c6-pre, p3, p4, p5)
    parts = split_by_depth(g, ops)
    assert [[o["conv"] for o in p] for p in parts] == [["c0", "c2", "c3"], ["c1"]]
    g, ops = _graph([(3, 256)] * 5)                                      # a shared head conv over five levels
    assert split_by_depth(g, ops) == [ops]
    g, ops = _graph([(1, 512), (1, 768)])                                # less than 2x apart
    assert split_by_depth(g, ops) == [ops]
    g, ops = _graph([(3, 64)])
    assert split_by_depth(g, ops) == [ops]
    monkeypatch.setenv("RNET_GROUP_SPLIT", "0")
    g, ops = _graph([(1, 2048), (1, 512)])
    assert split_by_depth(g, ops) == [ops]
