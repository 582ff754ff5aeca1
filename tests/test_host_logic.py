"""CPU-side tests of the product: state-dict contract, program compiler, C-ABI
exports, the host matcher (C++), sharding / record packing, and the N>1 path
over gloo.  No GPU, no compute kernels."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import decode_ref, hungarian_ref, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from rtpe import _native
    return _native


def test_abi_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "rtpe_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rtpe_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no prototypes found"
    lib = built.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(built.EXPORTS)
    assert lib.rtpe_version() == built.ABI_VERSION == 4
    assert lib.rtpe_device_count() >= 0


def test_state_dict_contract_of_product_module(w48_shapes):
    from rtpe.helpers import build_hrnet_w48_teacher
    m = build_hrnet_w48_teacher()
    sd = m.state_dict()
    assert list(sd.keys()) == ["1." + k for k in w48_shapes]          # same names, same order
    assert {k: tuple(v.shape) for k, v in sd.items()} == {"1." + k: v for k, v in w48_shapes.items()}
    # half wrapper: conv weights fp16, BatchNorm fp32 (fp16util.py:71-91 of the reference)
    assert sd["1.conv1.weight"].dtype == torch.float16
    assert sd["1.bn1.weight"].dtype == torch.float32 and sd["1.bn1.running_var"].dtype == torch.float32
    assert sd["1.final_layers.0.bias"].dtype == torch.float16
    # strict load of a checkpoint-shaped dict works and missing keys fail
    ck = {"1." + k: v for k, v in synth.make_state_dict(w48_shapes, 0, "W1").items()}
    m.load_state_dict(ck, strict=True)
    ck.pop("1.conv1.weight")
    with pytest.raises(RuntimeError):
        m.load_state_dict(ck, strict=True)
    assert m[1].__class__.__name__ == "PoseHigherResolutionNet"       # callers index [1]


def test_program_compiles_to_fused_ops(built):
    from rtpe.helpers import build_hrnet_w48_teacher
    prog = build_hrnet_w48_teacher()[1].compile_program()
    kinds = [op.kind for op in prog.ops]
    assert kinds.count(built.OP_STEM) == 1 and kinds.count(built.OP_DECONV) == 1
    # 302 Conv2d of the reference = 1 stem + 301 conv ops; 1 ConvTranspose2d; 23 fuse sums
    assert kinds.count(built.OP_CONV) == 301
    assert kinds.count(built.OP_FUSE) == 23
    assert prog.n_preds == 34 and prog.n_refined == 17
    assert len(prog.blob) > 127e6                         # 63.9 M fp16 weights + affine
    # lifetimes: at every op its input, residual and output live in different slots
    for op in prog.ops:
        if op.kind == built.OP_CONV and op.out_t >= 0 and op.out_t != op.in_t:
            slots = {prog.tensors[op.in_t].slot, prog.tensors[op.out_t].slot}
            assert len(slots) == 2
            if op.res_t >= 0:
                assert prog.tensors[op.res_t].slot != prog.tensors[op.out_t].slot


def test_sibling_downsampling_convs_are_neighbours_in_the_program(built):
    """the first convs of a fuse layer's downsampling chains from branch 0 (reference pose_higher_hrnet.py:213-230: 48 -> 96
    towards branch 1, 48 -> 48 towards branches 2 and 3) read the same map; HighResolutionModule.emit puts them next to each other
    on one lane, so that the executor can run them as one launch (csrc/conv48s2.hip, OpState::s2g): 4 pairs in stage 3, 2 triples
    in stage 4, nothing else changes (same op counts as the reference's module list)"""
    from rtpe.helpers import build_hrnet_w48_teacher
    prog = build_hrnet_w48_teacher()[1].compile_program()
    ops = list(prog.ops)
    is_s2 = lambda op: op.kind == built.OP_CONV and op.ksize == 3 and op.stride == 2 and op.cin == 48 and op.cout in (48, 96)
    runs, i = [], 0
    while i < len(ops):
        if is_s2(ops[i]):
            j = i + 1
            while j < len(ops) and is_s2(ops[j]) and (ops[j].in_t, ops[j].in_coff, ops[j].lane, ops[j].region) == \
                    (ops[i].in_t, ops[i].in_coff, ops[i].lane, ops[i].region):
                j += 1
            runs.append([ops[k].cout for k in range(i, j)])
            i = j
        else:
            i += 1
    groups = [r for r in runs if len(r) > 1]
    assert sorted(groups) == [[96, 48]] * 4 + [[96, 48, 48]] * 2, runs


def test_no_cpu_fallback(built):
    from rtpe.helpers import build_hrnet_w48_teacher
    from rtpe.third_party.pose_higher_hrnet import PoseHigherResolutionNet
    m = build_hrnet_w48_teacher()
    with pytest.raises(RuntimeError, match="HIP path only"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        m[1](torch.zeros(1, 3, 64, 64).half())
    assert PoseHigherResolutionNet(s2_modules=1).conv1.weight.dtype == torch.float32
    if not torch.cuda.is_available():
        from rtpe.third_party.group import HeatmapParser
        hp = HeatmapParser(17, 30, 0.1, 1.0, True, False)
        with pytest.raises(RuntimeError):
            hp.parse(torch.zeros(1, 17, 32, 32), torch.zeros(1, 17, 32, 32, 1))


def test_munkres_cpp_equals_restatement(built):
    from rtpe.third_party.group import py_max_match
    rng = np.random.default_rng(3)
    for _ in range(300):
        a, g = int(rng.integers(1, 31)), int(rng.integers(1, 31))
        # tie-heavy costs as group.py:66 builds them
        c = np.round(rng.random((a, g)) * 4) * 100 - rng.random((a, 1))
        if a > g:
            c = np.concatenate((c, np.full((a, a - g), 1e10)), 1)
        want = hungarian_ref.munkres_compute(c)
        got = [tuple(int(v) for v in p) for p in py_max_match(c)]
        assert got == want
    for n in (1, 2, 7):
        c = rng.random((n, n + 2))
        assert [tuple(p) for p in py_max_match(c)] == hungarian_ref.munkres_compute(c)


def test_munkres_cpp_equals_the_real_munkres_package(built, golden_dir):
    """the C++ solver (csrc/match_host.cpp) against Munkres().compute of the PyPI package the reference imports
    (group.py:19-23; vectors made on munkres 1.1.4 by tools/gen_golden.py): 700 mostly tie-heavy matrices"""
    from rtpe.third_party.group import py_max_match
    g = np.load(os.path.join(golden_dir, "munkres_vectors.npz"))
    assert int(g["real_package"]) == 1
    co = po = n = 0
    for (nr, nc), npairs in zip(g["shapes"], g["n_pairs"]):
        c = g["costs"][co:co + nr * nc].reshape(nr, nc)
        want = g["pairs"][po:po + npairs]
        got = py_max_match(c.copy())
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, want)
        co, po, n = co + nr * nc, po + npairs, n + 1
    assert n >= 500


def test_match_by_tag_cpp_equals_the_reference_on_real_munkres(built, golden_dir):
    """rtpe_match_by_tag against the reference's match_by_tag (group.py:26-97) run on the real package: quantised
    values / tags (equal-cost optima), people cap, ignore_too_much, use_detection_val off"""
    from rtpe.third_party.group import Params, match_by_tag
    g = np.load(os.path.join(golden_dir, "match_vectors.npz"))
    assert int(g["real_package"]) == 1
    for i in range(int(g["n_cases"])):
        mp, udv, itm = [int(v) for v in g["c%d_cfg" % i]]
        got = match_by_tag((g["c%d_tag" % i], g["c%d_loc" % i].astype(np.int64), g["c%d_val" % i]),
                           Params(17, mp, 0.1, 1.0, bool(udv), bool(itm)))
        want = g["c%d_ans" % i]
        assert got.shape == want.shape, i
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("name", ["p0", "p1", "p3", "p10", "p30", "p3_480", "p5_d2", "p40"])
def test_match_by_tag_cpp_matches_golden(built, golden_dir, name):
    from rtpe.third_party.group import HeatmapParser, Params, match_by_tag
    g = np.load(os.path.join(golden_dir, "decode_%s.npz" % name))
    params = Params(17, 30, 0.1, 1.0, True, False)
    got = match_by_tag((g["tag_k"][0], g["loc_k"][0], g["val_k"][0]), params)
    want = g["matched"]
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    hp = HeatmapParser(17, 30, 0.1, 1.0, True, False)
    np.testing.assert_array_equal(hp.match(g["tag_k"], g["loc_k"], g["val_k"])[0], want)


def test_match_by_tag_variants_against_oracle(built):
    from rtpe.third_party.group import Params, match_by_tag
    rng = np.random.default_rng(9)
    for trial in range(40):
        D = int(rng.choice([1, 1, 2, 9]))
        J, K = 17, 30
        val = np.sort(rng.random((J, K)).astype(np.float32) * (0.3 if trial % 3 else 1.0), axis=1)[:, ::-1]
        loc = rng.integers(0, 640, (J, K, 2)).astype(np.int64)
        tag = (rng.integers(0, 6, (J, K, D)) * 1.5 + rng.normal(0, 0.3, (J, K, D))).astype(np.float32)
        for kw in (dict(), dict(use_detection_val=False), dict(ignore_too_much=True, max_num_people=5),
                   dict(max_num_people=8)):
            a = dict(num_joints=J, max_num_people=30, detection_threshold=0.1, tag_threshold=1.0,
                     use_detection_val=True, ignore_too_much=False)
            a.update(kw)
            want = decode_ref.match_by_tag(tag, loc, np.ascontiguousarray(val), decode_ref.Params(**a))
            got = match_by_tag((tag, loc, np.ascontiguousarray(val)), Params(**a))
            assert got.shape == want.shape
            np.testing.assert_array_equal(got, want)


def test_sharding_and_records():
    from rtpe import engine
    assert [len(engine.shard_indices(100, r, 8)) for r in range(8)] == [13, 13, 13, 13, 12, 12, 12, 12]
    assert sum((engine.shard_indices(100, r, 8) for r in range(8)), []) == list(range(100))
    assert engine.RECORD_FLOATS * 4 == 8288                     # SURVEY.md section 8e record
    people = np.arange(2 * 17 * 4, dtype=np.float32).reshape(2, 17, 4)
    rec = engine.pack_records([7, 9], [(people, [0.5, 0.25]), (np.array([], np.float32), [])], "cpu")
    back = engine.unpack_records(rec)
    np.testing.assert_array_equal(back[7][0], people)
    np.testing.assert_array_equal(back[7][1], np.array([0.5, 0.25], np.float32))
    assert back[9][0].shape == (0, 17, 4)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtpe import engine
    ids = engine.shard_indices(5, rank, world)                  # 3 + 2 images
    res = [(np.full((1, 17, 4), float(i), np.float32), [float(i)]) for i in ids]
    rec = engine.pack_records(ids, res, "cpu")
    allrec = engine.all_gather_records(rec)
    sd = {"a": torch.arange(6.).reshape(2, 3) * (1 if rank == 0 else 0),
          "b": (torch.ones(4) * (3 if rank == 0 else 0)).half(),
          "n": torch.tensor(5 if rank == 0 else 0)}
    sd = engine.broadcast_state_dict(sd, 0, "cpu")
    out = engine.unpack_records(allrec)
    # equal shards (what bench.py runs): one collective, no count exchange
    ids2 = [rank * 2, rank * 2 + 1]
    rec2 = engine.pack_records(ids2, [(np.full((1, 17, 4), float(i), np.float32), [float(i)]) for i in ids2], "cpu")
    out2 = engine.unpack_records(engine.all_gather_records(rec2, equal_counts=True))
    assert sorted(out2) == [0, 1, 2, 3] and all(float(out2[i][1][0]) == float(i) for i in out2)
    q.put((rank, sorted(out.keys()), [float(out[i][0][0, 0, 0]) for i in sorted(out)],
           sd["a"].tolist(), sd["b"].tolist(), int(sd["n"])))
    dist.destroy_process_group()


def test_two_rank_gather_and_weight_broadcast_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, keys, vals, a, b, n in got:
        assert keys == [0, 1, 2, 3, 4] and vals == [0., 1., 2., 3., 4.]
        assert a == [[0., 1., 2.], [3., 4., 5.]] and b == [3.] * 4 and n == 5


def _one_rank_worker(port, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    from rtpe import engine
    calls = []
    real_bc, real_ag, real_agt = dist.broadcast, dist.all_gather, dist.all_gather_into_tensor
    dist.broadcast = lambda *a, **k: (calls.append("broadcast"), real_bc(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), real_ag(*a, **k))[1]
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append("all_gather_into_tensor"), real_agt(*a, **k))[1]
    sd = {"a": torch.arange(6.).reshape(2, 3), "b": torch.ones(4).half(), "n": torch.tensor(5)}
    assert engine.broadcast_state_dict(sd, 0, "cpu") is sd and not calls          # a world of one answers locally
    out = engine.broadcast_state_dict(sd, 0, "cpu", force_collective=True)
    assert calls == ["broadcast"] * 3 and all(torch.equal(out[k], sd[k]) for k in sd)
    del calls[:]
    names = ["%012d.jpg" % i for i in (3, 5, 8)]
    infer = lambda part: [(np.full((1, 17, 4), float(engine.image_id_of(n)), np.float32), [0.5]) for n in part]
    assert sorted(engine.run_sharded_list(names, infer, 2, "cpu")) == [3, 5, 8] and not calls
    got = engine.run_sharded_list(names, infer, 2, "cpu", force_collective=True)
    assert calls == ["all_gather", "all_gather_into_tensor"]
    assert sorted(got) == [3, 5, 8] and all(float(got[i][0][0, 0, 0]) == float(i) for i in got)
    q.put("ok")
    dist.destroy_process_group()


def test_force_collective_takes_the_collective_branch_with_one_rank():
    """the switch behind the one-GPU RCCL test (tests/test_gpu_parity.py): with it a process group of ONE rank
    issues the same collectives as a larger one; without it the world of one is answered locally"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(31500 + os.getpid() % 2000, q))
    p.start()
    assert q.get(timeout=120) == "ok"
    p.join(60)
    assert p.exitcode == 0


def test_student_state_dict_and_program(built, golden_dir):
    """config 5: AttentionStudent keeps the reference's parameter names/shapes (students.py:595-722,
    incl. the per-submodule ``load_state_dicts`` files) and compiles to one program"""
    from rtpe.students import AttentionStudent
    shapes = json.load(open(os.path.join(golden_dir, "student_shapes.json")))["shapes"]
    stu = AttentionStudent(None, "cpu", 100, 17, 1, True, None, False).eval()
    assert {k: list(v.shape) for k, v in stu.state_dict().items()} == shapes
    sd = synth.make_state_dict({k: tuple(v) for k, v in shapes.items()}, 3, "W1")
    stu.load_state_dict(sd, strict=True)
    assert stu.stem[1].conv1.weight.dtype == torch.float16          # half-wrapped stem
    assert stu.stem[1].bn1.weight.dtype == torch.float32 and stu.mid_stem[0].weight.dtype == torch.float32
    prog = stu.compile_program()
    kinds = [op.kind for op in prog.ops]
    assert kinds.count(built.OP_STEM) == 1 and kinds.count(built.OP_CAST) == 1
    assert kinds.count(built.OP_SE) == 5 and kinds.count(built.OP_CAM_COMBINE) == 5      # det_mid is never run
    assert kinds.count(built.OP_AVGPOOL) == 3 and kinds.count(built.OP_SIGMOID_ADD) == 1
    assert prog.n_preds == 1 and prog.n_refined == 18 and prog.outputs == [(1, 2), (18, 2)]
    # per-submodule files round-trip through load_state_dicts
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        pre = os.path.join(d, "x_")
        for name in ("mid_stem", "att_lo", "att_mid", "att_hi", "att_top"):
            torch.save({k: v + 1 for k, v in getattr(stu, name).state_dict().items()}, pre + name + ".statedict")
        before = stu.att_top[0].bias.clone()
        stu.load_state_dicts(pre)
        assert torch.equal(stu.att_top[0].bias, before + 1)


def test_preprocess_geometry(built):
    """row 8f-1: output size / centre / scale of resize_align_multi_scale (transforms.py:155-176) for the
    shapes SURVEY 0.5 works out by hand, product == oracle restatement, and the inverse map"""
    from oracle import preprocess_ref
    from rtpe.third_party import transforms
    for (h, w), want in (((480, 640), (896, 640)), ((555, 640), (768, 640)), ((640, 480), (640, 896)),
                         ((427, 640), (960, 640)), ((640, 640), (640, 640))):
        img = np.zeros((h, w, 3), np.uint8)
        size, center, scale = transforms.get_multi_scale_size(img, 640, 1, 1)
        size2, center2, scale2 = preprocess_ref.multi_scale_size(h, w, 640)
        assert size == want == size2
        np.testing.assert_array_equal(center, center2)
        np.testing.assert_allclose(scale, scale2, rtol=0, atol=0)
        m = transforms.get_affine_transform(center, scale, 0, size, inv=1)
        np.testing.assert_allclose(m, preprocess_ref.dst_to_src_matrix(center, scale, size), atol=1e-9)
        fwd = transforms.get_affine_transform(center, scale, 0, size)
        p = transforms.affine_transform(transforms.affine_transform([10., 20.], fwd), m)
        np.testing.assert_allclose(p, [10., 20.], atol=1e-9)
    # keypoints back to image coordinates (get_final_preds :195-202)
    size, center, scale = transforms.get_multi_scale_size(np.zeros((480, 640, 3), np.uint8), 640, 1, 1)
    person = np.array([[448., 320., 0.9, 1.0]])
    back = transforms.get_final_preds([[person]], center, scale, size)[0]
    np.testing.assert_allclose(back[0, :2], center, atol=1e-6)      # the centre of the warped image maps to the centre


def test_teacher_prediction_files(built, tmp_path):
    """row 8f-2: the npz schema of teacher_inference.py:86-90 and what dataloaders.py:149-154 reads back"""
    from rtpe import engine
    rng = np.random.default_rng(0)
    preds = rng.standard_normal((1, 34, 40, 56)).astype(np.float32)
    refined = rng.standard_normal((1, 17, 80, 112)).astype(np.float32)
    base = engine.teacher_prediction_path(str(tmp_path), "/data/coco/000000000139.jpg")
    assert base.endswith("000000000139.jpg_w48_predictions")
    engine.save_teacher_predictions(base, torch.from_numpy(preds), refined)
    npz = np.load(base + ".npz")
    assert sorted(npz.files) == ["embeddings", "heatmaps_order", "heatmaps_refined", "pred_heatmaps"]
    np.testing.assert_array_equal(npz["pred_heatmaps"], preds[0, :17])
    np.testing.assert_array_equal(npz["embeddings"], preds[0, 17:])
    np.testing.assert_array_equal(npz["heatmaps_refined"], refined[0])
    assert list(npz["heatmaps_order"]) == engine.HEATMAPS_ORDER and npz["heatmaps_order"].dtype.kind == "U"
    t_hms, t_ae = engine.load_teacher_predictions(base)
    assert t_hms.dtype == torch.float32 and tuple(t_hms.shape) == (17, 80, 112) and tuple(t_ae.shape) == (17, 40, 56)
    np.testing.assert_array_equal(t_ae.numpy(), preds[0, 17:])
    with pytest.raises(ValueError):
        engine.save_teacher_predictions(base, preds[:, :30], refined)


def test_checkpoint_loading_paths(built, w48_shapes, tmp_path):
    """the reference's loading entry points on a checkpoint file with the upstream key layout (``1.`` prefix):
    get_hrnet_w48_teacher (helpers.py:32-73), StemHRNet.load_pretrained / get_pretrained_stem
    (students.py:262-298) and AttentionStudent(hhrnet_statedict_path=...) (students.py:601-640)"""
    from rtpe.helpers import get_hrnet_w48_teacher
    from rtpe.students import AttentionStudent, get_pretrained_stem
    sd = synth.make_state_dict(w48_shapes, 0, "W1")
    path = str(tmp_path / "pose_higher_hrnet_w48_640.pth.tar")
    torch.save({"1." + k: v for k, v in sd.items()}, path)
    teacher = get_hrnet_w48_teacher(path)
    assert not teacher.training and teacher[1].conv1.weight.dtype == torch.float16
    assert torch.equal(teacher[1].bn1.running_mean, sd["bn1.running_mean"])
    assert torch.equal(teacher[1].conv1.weight, sd["conv1.weight"].half())
    stem = get_pretrained_stem(path, "cpu", True)
    assert torch.equal(stem[1].layer1[3].bn3.weight, sd["layer1.3.bn3.weight"])
    assert torch.equal(stem[1].layer1[0].downsample[0].weight, sd["layer1.0.downsample.0.weight"].half())
    stu = AttentionStudent(path, "cpu", 48, 17, 1, True, None, False)
    assert torch.equal(stu.stem[1].conv2.weight, sd["conv2.weight"].half())
    prog = stu.compile_program()
    assert prog.n_preds == 1 and prog.n_refined == 18


def test_stream_kernel_register_window_is_not_allocated(built, tmp_path):
    """conv_stream.hip keeps the residual rows that are in flight in v[224:255] and names those registers
    in its asm text; `amdgpu_num_vgpr(224)` is a target for the register allocator, not a hard limit, so a
    change that raises the pressure would let compiler temporaries land in the window (silent corruption).
    The device code of the built object must touch v224+ only through the three hand-written forms."""
    import re
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin/"
    import shutil
    # a copy: llvm-objcopy without an output file rewrites its input in place, and a test must not touch build products
    obj = shutil.copy(os.path.join(os.path.dirname(built.LIB_PATH), "build", "conv_stream.hip.o"), str(tmp_path / "s.o"))
    fat, co = str(tmp_path / "s.fatbin"), str(tmp_path / "s.co")
    subprocess.run([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, str(tmp_path / "s2.o")], check=True)
    subprocess.run([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + co], check=True)
    dis = subprocess.run([llvm + "llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
    mine = re.compile(r"global_load_dwordx4 v\[2\d\d:2\d\d\], v\d+, s\[|v_pk_add_f16 v\d+, v\d+, v2\d\d\b")
    n_window = 0
    in_v1 = False
    for line in dis.splitlines():
        label = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if label:
            in_v1 = "conv_stream_kernel" in label.group(1)
            continue
        if not in_v1:
            continue
        code = line.split("//")[0]
        hi = 0
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
            hi = max(hi, int(m.group(2)))
        for m in re.finditer(r"\bv(\d+)\b", code):
            hi = max(hi, int(m.group(1)))
        if hi >= 224:
            assert mine.search(code), "compiler-allocated register in the window: " + code.strip()
            n_window += 1
    assert n_window > 0


def _device_code(built, tmp_path, name):
    """disassembly of the gfx950 code object inside build/<name>.o (a copy: the test must not touch build products)"""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin/"
    obj = shutil.copy(os.path.join(os.path.dirname(built.LIB_PATH), "build", name + ".o"), str(tmp_path / (name + ".o")))
    fat, co = str(tmp_path / (name + ".fatbin")), str(tmp_path / (name + ".co"))
    subprocess.run([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, str(tmp_path / (name + ".2.o"))], check=True)
    subprocess.run([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + co], check=True)
    return subprocess.run([llvm + "llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def test_rounding_points_survive_the_compiler(built, tmp_path):
    """the half wrapper rounds a conv output and a BatchNorm output separately (two fp16 tensors in the reference).  The
    compiler fuses an fp32 fma whose result is cast to fp16 into v_fma_mixlo_f16 / v_fma_mixhi_f16, which rounds the exact
    a * b + c ONCE (one output in ~40,000 then differs by an fp16 step); every epilogue keeps the fma's result opaque to
    prevent that.  No conv kernel of the build may contain the fused form.  And the kernels of round 4 keep their design
    points: the fused stem runs conv1's fp32 chain on the matrix pipe (v_mfma_f32_16x16x4_f32), neither it nor the
    64-channel kernel spills registers to scratch memory"""
    import re
    for name in ("conv_mfma.hip", "conv_stream.hip", "conv_block.hip", "conv_direct.hip", "conv_pair.hip",
                 "stem_fused.hip", "conv64.hip", "deconv48.hip", "conv48s2.hip"):
        dis = _device_code(built, tmp_path, name)
        fused = re.findall(r"v_fma_mix(?:lo|hi)_f16", dis)
        assert not fused, "%s: %d fused fma + fp16 conversions" % (name, len(fused))
        if name in ("stem_fused.hip", "conv64.hip", "deconv48.hip", "conv48s2.hip"):
            # (the round-5 kernels keep 36 / 42 weight fragments per wave in registers: 220-236 of the 256 a wave may have)
            assert "scratch_" not in dis, name + ": registers spilled to scratch memory"
    dis = _device_code(built, tmp_path, "stem_fused.hip")
    assert dis.count("v_mfma_f32_16x16x4_f32") >= 4 * 7 and dis.count("v_mfma_f32_16x16x32_f16") >= 72
    dis = _device_code(built, tmp_path, "elementwise.hip")
    body = dis[dis.index("stem_kernelIDF16_"):] if "stem_kernelIDF16_" in dis else dis
    assert not re.findall(r"v_fma_mix(?:lo|hi)_f16", body.split("s_endpgm")[0]), "stem_kernel<half>: fused fma + conversion"


def test_compiled_engines_follow_the_weights(built, w48_shapes):
    """the executor holds a snapshot of the weights: in-place updates (load_state_dict on a sub-module,
    optimizer-style copy_, init_weights) must be noticed, a .to() that changes nothing must keep the engines, and a
    module that has engines can be pickled / deep-copied (CPU part: the bookkeeping, with a stand-in engine)"""
    import copy
    import pickle
    from rtpe.helpers import build_hrnet_w48_teacher
    from rtpe.students import AttentionStudent
    net = build_hrnet_w48_teacher()[1]
    assert len(net._fingerprint_tensors()["ts"]) == 1810
    sentinel = object()

    def arm():
        net.invalidate()
        net._engines[0] = sentinel
        net._fingerprint_tensors()

    def seen_within(calls):
        return any(net._weights_changed() for _ in range(calls))
    arm()
    assert not seen_within(8)                                   # nothing changed
    net.to("cpu").eval()                                        # same placement: engines are kept
    assert net._engines.get(0) is sentinel
    with torch.no_grad():
        net.final_layers[1].bias.copy_(net.final_layers[1].bias + 1)    # one tensor, in place
    assert seen_within(4)
    arm()
    with torch.no_grad():
        for p in net.parameters():                              # what an optimizer step does
            p.add_(0.0)
    assert net._weights_changed()                               # a bulk update: seen by the very next check
    arm()
    net.stage2[0].load_state_dict(net.stage2[0].state_dict())  # load_state_dict on a SUB-module
    assert 0 not in net._engines
    arm()
    net.init_weights()
    assert 0 not in net._engines
    arm()
    net.float()                                                 # new dtype: the packed weights are stale
    assert 0 not in net._engines
    net.half()
    # pickling / deep copies drop the native handles instead of failing on them
    net._engines[0] = sentinel
    clone = copy.deepcopy(net)
    assert clone._engines == {} and net._engines[0] is sentinel
    assert pickle.loads(pickle.dumps(net))._engines == {}
    net.invalidate()
    # training mode / autograd are refused (the executor is inference only)
    stu = AttentionStudent(None, "cpu", 48, 17, 1, True, None, False)
    assert stu.training
    with pytest.raises(RuntimeError):
        stu(torch.zeros(1, 3, 64, 64))


def test_host_thread_budget_is_divided_among_local_ranks(built, monkeypatch):
    """8 ranks x (16 matcher threads + 8 torch threads) is the burst that exhausts a container's CPU quota:
    the per-process budget is the allowed cores divided by LOCAL_WORLD_SIZE"""
    cores = len(os.sched_getaffinity(0))
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert built.host_threads(16) == min(16, cores)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert built.host_threads(16) == max(1, min(16, cores // 8))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1000")
    assert built.host_threads(16) == 1


def _list_worker(rank, world, port, names, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtpe import engine
    calls = []

    def infer(part):                                           # fake decode: n people that encode the image id
        calls.append(len(part))
        out = []
        for nm in part:
            i = engine.image_id_of(nm)
            n = i % 4
            out.append((np.full((n, 17, 4), float(i % 1000), np.float32) if n else np.array([], np.float32),
                        [float(i % 7)] * n))
        return out
    res = engine.run_sharded_list(names, infer, 5, "cpu")
    q.put((rank, calls, {k: (v[0].shape, float(v[0][0, 0, 0]) if len(v[0]) else None, v[1].tolist())
                         for k, v in res.items()}))
    dist.destroy_process_group()


def test_configs3_list_over_eight_ranks_gloo(golden_dir):
    """configs[3] control path without hardware: the 100 names of coco_minival2017_100.txt sharded over 8
    ranks (13,13,13,13,12,12,12,12), batches of 5 with short tails, variable-count all-gather, every image exactly
    once on every rank.  (The decode is faked; the collective code is the one bench.py --list runs on RCCL.)"""
    import torch.multiprocessing as mp
    from rtpe import engine
    names = [ln.strip() for ln in open(os.path.join(golden_dir, "coco_minival2017_100.txt")) if ln.strip()]
    assert len(names) == 100
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 23500 + os.getpid() % 2000
    procs = [ctx.Process(target=_list_worker, args=(r, 8, port, names, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ids = sorted(engine.image_id_of(n) for n in names)
    for rank, calls, res in got:
        assert calls == ([5, 5, 3] if rank < 4 else [5, 5, 2])
        assert sorted(res) == ids
        for i, (shape, first, scores) in res.items():
            assert shape == (i % 4, 17, 4) and scores == [float(i % 7)] * (i % 4)
            assert first is None or first == float(i % 1000)
    # single process: the same function without a process group
    out = engine.run_sharded_list(names[:7], lambda part: [(np.array([], np.float32), []) for _ in part], 3, "cpu")
    assert len(out) == 7
    with pytest.raises(ValueError):
        engine.run_sharded_list(names[:2] + names[:1], lambda part: [(np.array([], np.float32), [])] * len(part), 3, "cpu")


def test_student_steps_state_dict_and_program(built, golden_dir):
    """row 8f-3: AttentionStudentSteps keeps the reference's 340 parameter names / shapes (students.py:786-964) and
    compiles to one program with a second input"""
    from rtpe.students import AttentionStudentSteps
    shapes = json.load(open(os.path.join(golden_dir, "student_steps_shapes.json")))["shapes"]
    stu = AttentionStudentSteps(None, "cpu", 48, 17, 1, True, None, False).eval()
    assert {k: list(v.shape) for k, v in stu.state_dict().items()} == shapes and len(shapes) == 340
    prog = stu.compile_program(("att_divisor", 20.0))
    kinds = [op.kind for op in prog.ops]
    assert prog.has_aux and kinds.count(built.OP_AUX_PACK) == 1 and kinds.count(built.OP_RESIZE) == 1
    assert kinds.count(built.OP_GATE_MUL) == 1 and kinds.count(built.OP_SE) == 6 and kinds.count(built.OP_CAM_COMBINE) == 6
    assert prog.n_preds == 1 and prog.n_refined == 18 and prog.outputs == [(1, 2), (18, 2)]
    k5 = [op for op in prog.ops if op.kind == built.OP_CONV and op.ksize == 5]
    assert len(k5) == 2 and all(op.stride == 2 for op in k5)
    with pytest.raises(NotImplementedError):
        AttentionStudentSteps(None, "cpu", 50, 17, 1, True, None, False)


def test_alt_colour_space_restatement():
    """rgb2lab / rgb2hsv of the oracle (published scikit-image formulas) on known colours"""
    from oracle import student_ref
    rgb = np.array([[1., 1., 1.], [0., 0., 0.], [1., 0., 0.], [0., 1., 0.], [0., 0., 1.], [0.5, 0.5, 0.5]])
    lab = student_ref.rgb2lab(rgb)
    np.testing.assert_allclose(lab[0], [100.0, 0.0, 0.0], atol=2e-2)             # white (D65)
    np.testing.assert_allclose(lab[1], [0.0, 0.0, 0.0], atol=1e-9)
    np.testing.assert_allclose(lab[2], [53.24, 80.09, 67.20], atol=2e-2)         # sRGB red
    np.testing.assert_allclose(lab[4], [32.30, 79.19, -107.86], atol=2e-2)       # sRGB blue
    np.testing.assert_allclose(lab[5], [53.39, 0.0, 0.0], atol=2e-2)
    hsv = student_ref.rgb2hsv(rgb)
    np.testing.assert_allclose(hsv[2], [0.0, 1.0, 1.0])
    np.testing.assert_allclose(hsv[3], [1 / 3.0, 1.0, 1.0])
    np.testing.assert_allclose(hsv[4], [2 / 3.0, 1.0, 1.0])
    np.testing.assert_allclose(hsv[5], [0.0, 0.0, 0.5])


def test_autotune_cache_defaults_to_one_file_per_node_under_torchrun(monkeypatch):
    """RTPE_AUTOTUNE_CACHE unset: no file in a single process, one file per node and launch (temporary directory) when
    torch.distributed.run started several ranks on the node; an explicit value (also the empty string) wins."""
    from rtpe.third_party import pose_higher_hrnet as ph
    for k in ("RTPE_AUTOTUNE_CACHE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        monkeypatch.delenv(k, raising=False)
    assert ph._tuned_file() == ""
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert ph._tuned_file() == ""
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setenv("MASTER_PORT", "29512")
    a = ph._tuned_file()
    import tempfile
    assert a.startswith(tempfile.gettempdir()) and "29512" in a and a.endswith(".json")
    monkeypatch.setenv("MASTER_PORT", "29513")
    assert ph._tuned_file() != a                    # another launch, another file
    monkeypatch.setenv("RTPE_AUTOTUNE_CACHE", "")
    assert ph._tuned_file() == ""
    monkeypatch.setenv("RTPE_AUTOTUNE_CACHE", "/somewhere/t.json")
    assert ph._tuned_file() == "/somewhere/t.json"


def test_workspace_slot_is_per_thread():
    """the workspace slot of the forwards issued next (several forwards in flight use one each) belongs to the calling
    thread: a pipeline in another thread starts at slot 0 and does not see this thread's"""
    import threading
    from rtpe.third_party import pose_higher_hrnet as ph
    assert ph.set_workspace_slot(2) == 0
    try:
        seen = []
        t = threading.Thread(target=lambda: seen.append((ph._WS_SLOT[0], ph.set_workspace_slot(5))))
        t.start()
        t.join()
        assert seen == [(0, 0)] and ph._WS_SLOT[0] == 2
    finally:
        assert ph.set_workspace_slot(0) == 2


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without WORLD_SIZE in the environment (the form the driver uses) must run TWO ranks:
    bench.py starts `torch.distributed.run` as a child before anything touches a GPU and passes rank 0's JSON line
    through.  Here on the CPU: --rehearsal (control path only: process group, weight broadcast, shards, record gather,
    barrier + max-over-ranks timing; no GPU work) over gloo."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rehearsal",
                          "--steps", "3", "--warmup", "1", "--batch", "4"], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rehearsal"] is True and out["config"]["records_gathered"] == 8
    assert out["steps"] == 3 and out["warmup"] == 1 and out["value"] is None
    # and the single-rank form stays in this process (no child, no process group)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearsal", "--steps", "2",
                          "--warmup", "0", "--batch", "3"], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["records_gathered"] == 3 and "starting" not in res.stderr
