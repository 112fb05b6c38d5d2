"""ctypes binding of libvcgan_hip.so (include/vc_cyclegan.h).

The product path has NO fallback: if the library is missing or fails to load,
importing this module raises.  (The build container has no GPU but does have
libamdhip64, so loading and symbol checks work there; launching does not.)
"""
import ctypes as C
import os

# torch bundles its own libamdhip64; load it FIRST so libvcgan_hip.so (linked against the SONAME
# libamdhip64.so.7) binds to that already-loaded runtime.  The other order leaves two HIP runtimes in
# the process and the second one to initialise reports "no ROCm-capable device".
import torch  # noqa: F401,E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'csrc', 'libvcgan_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        'libvcgan_hip.so not built: run `python -c "import __graft_entry__ as g; g.build()"` '
        '(hipcc --offload-arch=gfx950).  There is no CPU or eager fallback for the hot path.')


def _expected_build_id():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_vcgan_build_id', os.path.join(HERE, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, mod.build_id(os.environ.get('VC_EXTRA_FLAGS', '').split())


ABI_VERSION = 116
lib = C.CDLL(LIB_PATH)
lib.vc_build_id.restype = C.c_char_p
BUILD_ID = lib.vc_build_id().decode()
_bmod, _want = _expected_build_id()
if BUILD_ID != _want and not os.environ.get('VC_ALLOW_STALE_LIB'):
    # a library built from other sources than the ones next to it must never be tested or benched: the import is REFUSED (never an
    # implicit rebuild -- the caller runs __graft_entry__.build()).  (VC_ALLOW_STALE_LIB=1: instrumented builds of tools/.)
    if os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        raise ImportError('libvcgan_hip.so was built from other sources (build id %s, tree %s): run '
                          '`python -c "import __graft_entry__ as g; g.build()"` before importing the package' % (BUILD_ID, _want))
    raise ImportError('libvcgan_hip.so is stale (build id %s, tree %s) and no hipcc is available to rebuild it' % (BUILD_ID, _want))

i32, i64, f32, vp = C.c_int, C.c_int64, C.c_float, C.c_void_p


class ParamRec(C.Structure):
    _fields_ = [('name', C.c_char * 96), ('offset', C.c_int64), ('ndim', C.c_int32),
                ('shape', C.c_int32 * 4), ('ld', C.c_int32)]


# (name, restype, argtypes) -- every symbol declared in include/vc_cyclegan.h
SIGNATURES = [
    ('vc_version', i32, []),
    ('vc_thread_is_product_run', i32, []),
    ('vc_build_id', C.c_char_p, []),
    ('vc_last_error', C.c_char_p, []),
    ('vc_crc32c', C.c_uint32, [vp, C.c_uint64, C.c_uint32]),
    ('vc_set_math_mode', i32, [i32]),
    ('vc_get_math_mode', i32, []),
    ('vc_release_workspaces', i32, []),
    ('vc_set_deterministic', i32, [i32]),
    ('vc_get_deterministic', i32, []),
    ('vc_tune_igemm_override', i32, [i32, i32, i32]),
    ('vc_tune_wgrad_override', i32, [i32, i32, i32]),
    ('vc_tune_wgrad_store', i32, [i32]),
    ('vc_tune_rowskip', i32, [i32]),
    ('vc_tune_conv_bf16_override', i32, [i32, i32, i32]),
    ('vc_tune_conv_bf16_spec', i32, [i32]),
    ('vc_tune_bf16_pixel_octets', i32, [i32]),
    ('vc_tune_conv_bf16_keys', i32, [vp, i32]),
    ('vc_tune_last_igemm_config', i32, [C.POINTER(C.c_int)]),
    ('vc_tune_last_wgrad_config', i32, [C.POINTER(C.c_int)]),
    ('vc_tune_query_igemm', i32, [i32, i32, i32, i32, i32, C.POINTER(C.c_int)]),
    ('vc_tune_query_wgrad', i32, [i32, i32, i32, i32, i32, C.POINTER(C.c_int)]),
    ('vc_prof_enable', i32, [i32]),
    ('vc_prof_collect', i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    ('vc_prof_records', i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), i32]),
    ('vc_param_layout', i32, [i32, i32, C.POINTER(ParamRec), i32, C.POINTER(i64)]),
    ('vc_conv2d_fwd', i32, [vp, i64, vp, i64, vp, i64, vp, i64] + [i32] * 10 + [vp]),
    ('vc_conv2d_dgrad', i32, [vp, i64, vp, i64, vp, i64, vp, i64, i32] + [i32] * 10 + [vp]),
    ('vc_conv2d_wgrad', i32, [vp, i64, vp, i64, vp, i64] + [i32] * 10 + [vp]),
    ('vc_gated_conv1d_glu_fwd', i32, [vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    ('vc_gated_conv1d_glu_image_bytes', i64, [i32, i32, i32]),
    ('vc_gated_conv1d_glu_prepack', i32, [vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    ('vc_gated_conv1d_glu_fwd_packed', i32, [vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp]),
    ('vc_plan_num_step_graphs', i32, [vp]),
    ('vc_tune_glu', i32, [i32]),
    ('vc_tune_c1_dgrad_gather', i32, [i32]),
    ('vc_tune_layer_updates', i32, [i32]),
    ('vc_tune_winograd', i32, [i32]),
    ('vc_testing_wino_gemms', C.c_longlong, [i32]),
    ('vc_testing_wino_saved_flops', C.c_double, [i32, i32]),
    ('vc_testing_hooks_active', i32, []),
    ('vc_testing_describe_step', i64, [vp, i32, i32, C.c_char_p, i64]),
    ('vc_testing_poison_internal', i32, [vp, vp, i32]),
    ('vc_prof_glu_copy_roof', i32, [vp, vp, vp, i32, i32, i32, vp]),
    ('vc_bias_grad', i32, [vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    ('vc_instnorm_glu_fwd', i32, [i32, i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    ('vc_instnorm_glu_bwd', i32, [i32, i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64,
                                  i32, i32, i32, i32, vp]),
    ('vc_disc_head_fwd', i32, [vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    ('vc_disc_head_bwd', i32, [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, vp]),
    ('vc_crop_batch', i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    ('vc_l1_loss', i32, [vp, vp, i64, vp, vp, vp]),
    ('vc_l2_loss', i32, [vp, vp, i64, vp, vp, vp]),
    ('vc_adam_tf_update', i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp]),
    ('vc_plan_create', i32, [i32, i32, C.POINTER(vp)]),
    ('vc_plan_create_inference', i32, [i32, i32, C.POINTER(vp)]),
    ('vc_plan_destroy', None, [vp]),
    ('vc_plan_set_math_mode', i32, [vp, i32]),
    ('vc_plan_get_math_mode', i32, [vp]),
    ('vc_plan_set_deterministic', i32, [vp, i32]),
    ('vc_plan_get_deterministic', i32, [vp]),
    ('vc_generator_save_floats', i64, [vp, i32]),
    ('vc_discriminator_save_floats', i64, [vp, i32]),
    ('vc_scratch_floats', i64, [vp, i32]),
    ('vc_generator_forward', i32, [vp, vp, i64, vp, i64, vp, i64, vp, i32, i32, vp]),
    ('vc_generator_backward', i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, vp, i64, i32, i32, vp]),
    ('vc_discriminator_out_elems', i32, [vp]),
    ('vc_discriminator_forward', i32, [vp, vp, i64, vp, i64, vp, i64, vp, i32, i32, vp]),
    ('vc_discriminator_backward', i32, [vp, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, vp, i64, i32, i32, vp]),
    ('vc_flat_param_floats', i64, [vp]),
    ('vc_flat_offset', i64, [vp, i32]),
    ('vc_step_workspace_floats', i64, [vp, i32]),
    ('vc_cyclegan_step_grads', i32, [vp, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp, vp, i64, vp]),
    ('vc_plan_set_overlap', i32, [vp, i32]),
    ('vc_plan_set_exchange_mode', i32, [vp, i32]),
    ('vc_plan_get_exchange_mode', i32, [vp]),
    ('vc_step_wait_d_grads', i32, [vp, vp]),
    ('vc_step_num_g_buckets', i32, [vp]),
    ('vc_step_g_bucket', i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
    ('vc_step_wait_g_bucket', i32, [vp, i32, vp]),
    ('vc_train_step_graph', i32, [vp, vp, vp, vp, vp, vp, vp, i32, f32, f32, f32, f32, f32, f32, f32, vp, vp, vp, vp, i64, vp]),
    ('vc_comm_unique_id', i32, [vp]),
    ('vc_comm_init', i32, [i32, i32, vp, C.POINTER(vp)]),
    ('vc_comm_destroy', i32, [vp]),
    ('vc_comm_world', i32, [vp]),
    ('vc_comm_collectives', i64, [vp]),
    ('vc_allreduce_f32', i32, [vp, vp, i64, vp]),
    ('vc_step_exchange_and_update', i32, [vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, f32, vp]),
]
for _n, _r, _a in SIGNATURES:
    _f = getattr(lib, _n)          # AttributeError here == header/library mismatch
    _f.restype = _r
    _f.argtypes = _a


if lib.vc_version() != ABI_VERSION:
    raise ImportError('libvcgan_hip.so speaks ABI revision %d, this binding was written against %d' % (lib.vc_version(), ABI_VERSION))


class VcError(RuntimeError):
    pass


def check(rc, what=''):
    if rc != 0:
        msg = lib.vc_last_error()
        raise VcError('%s failed (rc=%d): %s' % (what or 'libvcgan_hip call', rc, msg.decode() if msg else ''))


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())
