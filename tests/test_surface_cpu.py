"""CPU: the §8(b) Python surface that needs no GPU — command-line flags, config -> graph through the sub-builders,
every shipped reference config (when /root/reference is mounted), executor helpers, input-pipeline sharding for the
one-process-per-GPU layout, and the ctypes structs against the C header's own layout."""
import copy
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

REF_CONFIGS = "/root/reference/configs"


def test_flags_parse_like_absl():
    from retinanet.__main__ import define_flags
    from retinanet.flags import FlagError
    f = define_flags().parse(["--config_path=a.json", "--model_dir", "out", "--debug", "--norun_evaluation",
                              "--is_multi_host=true", "--global_seed=7", "--log_dir", "logs"])
    assert (f.config_path, f.model_dir, f.debug, f.run_evaluation, f.is_multi_host, f.global_seed, f.log_dir) == \
        ("a.json", "out", True, False, True, 7, "logs")
    assert f.resume_from is None and f.xla is False and f.run_continuous_evaluation is False
    with pytest.raises(FlagError):
        define_flags().parse(["--no_such_flag"])
    from retinanet.export import define_flags as export_flags
    e = export_flags().parse(["--config_path", "c.json", "--mode", "tf", "--export_saved_model"])
    assert e.mode == "tf" and e.export_saved_model and e.export_dir == "export" and e.checkpoint_name == "latest"
    with pytest.raises(FlagError):
        export_flags().parse(["--config_path", "c.json", "--mode", "tflite"])     # not in the enum
    with pytest.raises(FlagError):
        export_flags().parse(["--mode", "tf"])                                    # config_path is required


def test_sub_builders_compose_the_reference_model(params):
    """model/builder.py:36-106 step by step through build_backbone / build_neck / build_detection_heads."""
    from retinanet.model.backbone import build_backbone
    from retinanet.model.graph import build_retinanet_graph, graph_input
    from retinanet.model.head import build_detection_heads
    from retinanet.model.layers.balance_features import BalanceFeatures
    from retinanet.model.neck import build_neck
    from retinanet.model.utils import get_activation_op, get_normalization_op
    arch = params.architecture
    images = graph_input([640, 640, 3])
    act = get_activation_op(arch.activation.type)
    feats = build_backbone([640, 640, 3], arch.backbone, arch.batch_norm)(images)
    assert {k: v.shape for k, v in feats.items()} == {"2": (None, 160, 160, 256), "3": (None, 80, 80, 512),
                                                     "4": (None, 40, 40, 1024), "5": (None, 20, 20, 2048)}
    feats = build_neck(arch.feature_fusion, arch.conv_2d, arch.batch_norm, act)(feats)
    assert [feats[str(l)].shape for l in range(3, 8)] == [(None, s, s, 256) for s in (80, 40, 20, 10, 5)]
    feats = BalanceFeatures(3, 7, 4)(feats)
    box_head, class_head = build_detection_heads(arch.head, 3, 7, arch.conv_2d, arch.batch_norm, act)
    assert box_head(feats)["3"].shape == (None, 80, 80, 36) and class_head(feats)["7"].shape == (None, 5, 5, 720)
    g = images.graph
    n_params = sum(int(np.prod(v["shape"])) for v in g.var_specs.values() if v.get("trainable", True))
    assert n_params == 34_389_556
    assert len([k for k in g.var_specs if g.var_specs[k].get("trainable", True)]) == 295
    assert g.var_specs["class-head/class-head-prediction-conv2d/bias"]["value"] == pytest.approx(-4.59511985)
    assert sorted(g.var_specs) == sorted(build_retinanet_graph(params).var_specs)
    # error behaviour of the reference's builders
    with pytest.raises(ValueError):
        build_neck(arch.feature_fusion, arch.conv_2d, arch.batch_norm, None)
    with pytest.raises(ValueError):
        build_detection_heads(arch.head, 3, 7, arch.conv_2d, arch.batch_norm, None)
    bad = copy.deepcopy(arch.feature_fusion)
    bad.type = "bifpn"
    with pytest.raises(ValueError, match="FPN not implemented"):
        build_neck(bad, arch.conv_2d, arch.batch_norm, act)
    bad_bb = copy.deepcopy(arch.backbone)
    bad_bb.type = "vgg"
    with pytest.raises(ValueError, match="backbone not implemented"):
        build_backbone([640, 640, 3], bad_bb, arch.batch_norm)
    with pytest.raises(AssertionError):
        BalanceFeatures(3, 7, 9)
    with pytest.raises(ValueError):
        get_activation_op("gelu")
    assert get_normalization_op(use_sync=True, num_replicas=1)["sync"] is False      # model/utils.py:10-12
    assert get_normalization_op(use_sync=True, num_replicas=8)["kind"] == "sync_batch_normalization"


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference mount absent (GPU box)")
def test_every_shipped_config_loads_and_builds_or_says_why():
    from retinanet.cfg import Config
    from retinanet.model.graph import build_retinanet_graph
    from retinanet.optimizers import build_optimizer
    paths = sorted(glob.glob(os.path.join(REF_CONFIGS, "**", "*.json"), recursive=True))
    assert len(paths) >= 15
    built, refused = [], []
    for path in paths:
        p = Config(path).params
        opt = build_optimizer(p.training.optimizer, p.training.train_steps, precision=p.floatx.precision)
        assert opt.lr(0) > 0 and opt.clipnorm == 10.0
        kind = p.architecture.backbone.type.lower()
        try:
            g = build_retinanet_graph(p)
        except NotImplementedError as e:
            assert "mobiledet" in kind or "lite" in kind, (path, e)      # documented out-of-scope backbones
            refused.append(os.path.basename(path))
            continue
        assert "resnet" in kind or kind.startswith("efficientnet-b"), path
        levels = range(int(p.architecture.feature_fusion.min_level), int(p.architecture.feature_fusion.max_level) + 1)
        assert sorted(g.outputs["class-predictions"]) == [str(l) for l in levels]
        H = p.input.input_shape[0]
        A, K = p.architecture.head.num_anchors, p.architecture.head.num_classes
        assert g.tensors[g.outputs["class-predictions"]["3"]][:3] == (H // 8, H // 8, A * K)
        built.append(os.path.basename(path))
    assert len(built) >= 7 and len(refused) >= 5, (built, refused)


def test_executor_helpers_known_answers():
    from retinanet.executor import AverageMeter, InflectionDetector, format_eta
    assert format_eta(3725) == "01h 02m 05s"
    m = AverageMeter(momentum=0.5)
    for v in range(12):
        m.accumulate(float(v))
    assert m.averaged_value == pytest.approx((9 * 0.5 + 10 * 0.5) * 0.5 + 11 * 0.5)   # first 10 values overwrite
    d = InflectionDetector("l2", threshold=0.05, skip_steps=5)
    assert not any(d.is_value_anomalous(1.0 - 0.01 * i) for i in range(20))            # a straight line: no inflection
    vals = [1.0 - 0.01 * i for i in range(10)] + [5.0, 5.0, 5.0]
    d.reset()
    assert any(d.is_value_anomalous(v) for v in vals)


def test_input_pipeline_ranks_see_disjoint_records_on_one_host(tmp_path, params):
    """one process per GPU means every rank is an input pipeline of its own, with or without
    `is_multi_host` — files sharded, batch divided, draws seeded per rank."""
    from test_tfrecord_cpu import _write_dataset
    from retinanet.dataloader.input_pipeline import InputContext, InputPipeline
    from retinanet.dataloader.tfrecord_parser import parse_example
    _write_dataset(tmp_path, 16, 8)
    p = copy.deepcopy(params)
    p.dataloader_params["tfrecords"] = {"train": str(tmp_path / "train-*"), "val": str(tmp_path / "train-*")}
    p.training.batch_size = {"train": 8, "val": 8}
    pipe = InputPipeline("val", p, is_multi_host=False, num_replicas=2)
    ids = []
    for rank in range(2):
        mine = [parse_example(r, decode=False)["image_id"] for r in pipe._records(InputContext(2, rank, 2))]
        assert len(mine) == 8
        ids.append(set(mine))
    assert not (ids[0] & ids[1]) and (ids[0] | ids[1]) == set(range(1000, 1016))
    assert InputContext(2, 1, 2).get_per_replica_batch_size(8) == 4
    with pytest.raises(ValueError):
        list(pipe._files(InputContext(32, 0, 32)))        # fewer files than pipelines


def test_ctypes_structs_match_the_c_header(tmp_path):
    """sizeof / offsetof of every struct the binding mirrors, taken from the C compiler's view of include/rnet_hip.h"""
    from retinanet import _C
    structs = {"rn_conv_segment": (_C.ConvSegment, ["x", "bias", "w_terms", "w_pair", "Cout", "bn_partial"]),
               "rn_launch_opts": (_C.LaunchOpts, ["conv_tile", "reserved_cus", "wgrad_kernel", "ablate"]),
               "rn_conv_problem": (_C.ConvProblem, ["seg", "num_segments", "out_dtype", "opts"]),
               "rn_wgrad_segment": (_C.WgradSegment, ["dy_pix_stride", "x_pix_stride"]),
               "rn_wgrad_problem": (_C.WgradProblem, ["seg", "opts"]),
               "rn_dw_segment": (_C.DwSegment, ["residual", "Wo"]),
               "rn_dw_problem": (_C.DwProblem, ["seg"]),
               "rn_bn_segment": (_C.BnSegment, ["sample_scale", "ext_chunks", "P"]),
               "rn_bn_problem": (_C.BnProblem, ["seg", "count_scale"]),
               "rn_dgrad_pack": (_C.DgradPack, ["Cout_pad"]),
               "rn_bottleneck64_problem": (_C.Bottleneck64Problem, ["w_packed", "affine", "N", "Cx", "opts"]),
               "rn_example_info": (_C.ExampleInfo, ["n_classes"])}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rnet_hip.h"', "int main(void) {"]
    for name, (_, fields) in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fld in fields:
            lines.append(f'  printf("{name}.{fld} %zu\\n", offsetof({name}, {fld}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    want = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    import ctypes
    for name, (cls, fields) in structs.items():
        assert ctypes.sizeof(cls) == int(want[name]), name
        for fld in fields:
            assert getattr(cls, fld).offset == int(want[f"{name}.{fld}"]), (name, fld)
