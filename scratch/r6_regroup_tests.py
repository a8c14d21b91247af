"""One-off (round 6): regroup tests/test_hip_parity*.py (organised by ROUND) into tests/test_gpu_<component>.py
(organised by COMPONENT), helpers into tests/gpu_common.py; every test keeps its body verbatim and gets its round as a docstring tag."""
import ast
import os
import re

T = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")
files = [("test_hip_parity.py", 1), ("test_hip_parity_r2.py", 2), ("test_hip_parity_r3.py", 3), ("test_hip_parity_r4.py", 4),
         ("test_hip_parity_r5.py", 5)]
rules = [("dist", ["rccl", "dist", "gloo", "rank", "_dp_", "launcher", "data_parallel", "replicas", "exchange", "bucket"]),
         ("step_runtime", ["record", "replay", "captur", "graph", "overlap", "handoff", "arena", "auto_record", "recording", "two_models", "eager",
                           "update_auto", "train_loop", "amax_pool", "adamw"]),
         ("fft_dc", ["fft", "rss", "sens_", "dc_rows", "cascade_boundary", "varnetblock", "dc_weight"]),
         ("warp_loss", ["warp", "ssim", "lncc", "gradient_loss", "grid_sample", "augment", "metric", "smooth", "loss"]),
         ("norm", ["norm", "act_bwd", "bn_", "plane_stats", "unshuffl", "avgpool", "elementwise", "window_copy", "plane_activation"]),
         ("conv", ["conv", "tconv", "gemm", "wgrad", "f16x2", "fp8", "bf16", "precision", "splitk", "direct", "stream", "transposed", "kernels_repeat",
                   "weight_gradient"]),
         ("e2e", [""])]
override = {"test_lncc_through_warp_and_fft_autograd_vs_reference": "warp_loss", "test_reference_smoke_idiom_matches_direct_chain": "e2e",
            "test_loss_all_backward_matches_update_chain_bitwise": "e2e", "test_full_rec_step_gradients_elementwise_on_shipped_kernels": "e2e",
            "test_full_rec_step_with_bf16x3_convs": "e2e", "test_weight_gradient_handoff_batch_size_changes_nothing": "step_runtime",
            "test_conv_precision_modes_e2e_psnr": "e2e", "test_fp8_mode_e2e_psnr_and_train_step": "e2e", "test_mixed_backward_precision_full_320": "e2e",
            "test_narrow_precision_psnr_on_trained_like_weights": "e2e", "test_normunet_pad_golden": "e2e", "test_unet_reflect_pad_golden": "e2e",
            "test_varnet_pad_golden": "e2e", "test_full_rec_step_gradients_vs_golden": "e2e", "test_varnet_backward_vs_golden_grads": "e2e",
            "test_unet_backward_vs_oracle_autograd": "e2e", "test_normunet_backward_vs_oracle_autograd": "e2e",
            "test_alignment_backward_in_eval_mode_vs_oracle_autograd": "e2e", "test_alignment_layers_golden": "e2e"}
renames = {2: {"_digest_errors": "_digest_errors_r2"}, 3: {"_digest_errors": "_digest_errors_r3"}, 4: {"_model": "_model_r4"}, 5: {"_model": "_model_r5"}}
titles = {"fft_dc": "FFT family and the fused cascade boundary (fft2 / ifft2 / rss, sens_reduce / sens_expand, dc_rows; SURVEY 8 rows a1-a5)",
          "conv": "convolution kernels: 3x3 / 1x1 / transposed forward, data and weight gradients, every operand format (rows a9, N1)",
          "norm": "normalisation, activation and the element-wise materialisers (rows a8, a9, a11)",
          "warp_loss": "warp / grid_sample, SSIM / LNCC / smoothness losses and their backward, augmentation, metrics (rows a12-a15, f3, f4)",
          "e2e": "end-to-end: VarNet / NormUnet / alignment net against the golden fixtures, training steps, narrow-precision PSNR "
                 "(rows a6, a7, a10, a16, a17)",
          "step_runtime": "the step's runtime: recorded / captured steps, stream overlap, arenas, determinism, the fused optimiser (rows f1, d)",
          "dist": "multi-rank paths that need the GPU (gloo on one GPU, one-rank RCCL) (row e)"}
imports, helpers, seen_helper = [], [], {}
tests = {k: [] for k, _ in rules}
for fname, rnd in files:
    src = open(os.path.join(T, fname)).read()
    lines = src.splitlines()
    tree = ast.parse(src)
    ren = renames.get(rnd, {})

    def seg(node):
        lo = min([node.lineno] + [d.lineno for d in getattr(node, "decorator_list", [])])
        while lo - 2 >= 0 and lines[lo - 2].lstrip().startswith("#"):      # the comment block right above
            lo -= 1
        text = "\n".join(lines[lo - 1:node.end_lineno])
        for a, b in ren.items():
            text = re.sub(r"(?<![\w.])" + re.escape(a) + r"\b", b, text)
        return text

    for node in tree.body:
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            t = ast.get_source_segment(src, node)
            if t not in imports:
                imports.append(t)
        elif isinstance(node, ast.FunctionDef) and node.name.startswith("test_"):
            comp = override.get(node.name)
            if comp is None:
                comp = next(c for c, keys in rules if any(k in node.name for k in keys))
            text = seg(node)
            doc = ast.get_docstring(node, clean=False)
            if doc is not None:
                text = text.replace('"""' + doc, '"""[round %d] ' % rnd + doc, 1)
            else:
                body_line = lines[node.body[0].lineno - 1]
                text = text.replace("\n" + body_line, '\n    """[round %d]"""\n' % rnd + body_line, 1)
            tests[comp].append(text)
        elif isinstance(node, (ast.FunctionDef, ast.Assign)):
            name = node.name if isinstance(node, ast.FunctionDef) else node.targets[0].id
            if name in ("S", "pytestmark", "DEV", "g"):
                continue
            name = ren.get(name, name)
            text = seg(node)
            if name in seen_helper:
                assert seen_helper[name] == ast.dump(node), name
                continue
            seen_helper[name] = ast.dump(node)
            helpers.append(text)
imp = "\n".join(i for i in imports if "conftest" not in i)
common = '''"""Helpers shared by the component-grouped GPU test files (tests/test_gpu_*.py): the module namespace fixture, tensor movers,
model builders, digests, worker functions of the multi-process tests.  (Round 6: the tests were regrouped from one file per ROUND
into one file per COMPONENT; a helper that two rounds defined differently keeps both forms with a round suffix.)"""
%s
from conftest import as_t, cplx, philox, rel_err, load_golden  # noqa: F401

DEV = "cuda:0"


@pytest.fixture(scope="module")
def S():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spatialalignmentnetwork_amd import (ops, synth, varnet, cross, unet, signal_utils, ssimloss, lnccloss, masks, model, basemodel,
                                             autograd, _lib)
    from oracle import cpu_ref as O

    class NS:
        pass

    ns = NS()
    ns.ops, ns.synth, ns.varnet, ns.cross, ns.unet, ns.sig, ns.ssim, ns.lncc = ops, synth, varnet, cross, unet, signal_utils, ssimloss, lnccloss
    ns.masks, ns.model, ns.base, ns.O, ns.autograd, ns.lib = masks, model, basemodel, O, autograd, _lib
    return ns


def g(t):
    return t.to(DEV).contiguous()


%s
''' % (imp, "\n\n\n".join(helpers))
open(os.path.join(T, "gpu_common.py"), "w").write(common)
names = ["S", "g", "DEV"] + list(seen_helper)
for comp, items in tests.items():
    if not items:
        continue
    body = '''"""GPU parity tests (through the C ABI), component: %s.
Every test carries the round it was written in as a docstring tag; tolerances are written next to the comparisons."""
%s
from conftest import as_t, cplx, philox, rel_err, load_golden  # noqa: F401
from gpu_common import (%s)  # noqa: F401

pytestmark = pytest.mark.gpu


%s
''' % (titles[comp], imp, ", ".join(names), "\n\n\n".join(items))
    open(os.path.join(T, "test_gpu_%s.py" % comp), "w").write(body)
    print(comp, len(items))
for fname, _ in files:
    os.remove(os.path.join(T, fname))
