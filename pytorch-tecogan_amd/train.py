"""Drop-in for the reference's code/train.py: FRVSR_Train / TecoGAN with the same signature, side effects (both
modules' parameters / BN buffers and both optimisers updated in place) and the same `Network` result tuple
(code/train.py:354-377).  The arithmetic runs in step.TecoGANStep on HIP kernels."""
import collections

import torch

from . import _lib as L
from . import parallel
from . import tuning
from .models import compute_dtype
from .step import TecoGANStep

Network = collections.namedtuple(
    "Network",
    "gen_output, learning_rate, update_list, update_list_name, update_list_avg, global_step, d_loss, gen_loss, "
    "fnet_loss ,tb, target")

_STEPS = {}



def _bind_optimizer(opt, module):
    """makes torch.optim.Adam's state tensors views of the engine's flat moment buffers, so optimizer.state_dict()
    (checkpoint ABI, main.py:308-317) stays meaningful although the update is one fused HIP launch."""
    flat = module.flat_params()
    tag = (id(flat), id(opt))
    if getattr(opt, "_tg_bound", None) == tag:
        # optimizer.load_state_dict() after the first step swaps in fresh state tensors without touching the tag:
        # trust the binding only while the state still IS the flat moment buffer
        name0, p0 = next(iter(module.named_parameters()))
        st0 = opt.state.get(p0)
        if st0 and torch.is_tensor(st0.get("exp_avg")) and st0["exp_avg"].data_ptr() == flat.view(flat.m, name0).data_ptr() \
                and st0.get("step") is opt._tg_step:
            return
    shared_step = None
    for name, p in module.named_parameters():
        st = opt.state.get(p, None)
        m, v = flat.view(flat.m, name), flat.view(flat.v, name)
        if st and "exp_avg" in st and st["exp_avg"].data_ptr() != m.data_ptr():
            m.copy_(st["exp_avg"].to(m.device))
            v.copy_(st["exp_avg_sq"].to(v.device))
            if shared_step is None:
                shared_step = torch.tensor(float(st["step"]))
        if shared_step is None:
            shared_step = torch.tensor(0.0)
        opt.state[p] = {"step": shared_step, "exp_avg": m, "exp_avg_sq": v}
    opt._tg_bound = tag
    opt._tg_step = shared_step


_PENDING_SCALER = None


def loss_scaler_state():
    """GradScaler.state_dict()-style {'scale', 'growth_tracker'} of the live step's fp16 loss scaler (None outside fp16 mode):
    what main.py stores next to the optimiser state as the extra checkpoint key `tg_scaler`."""
    for st in _STEPS.values():
        return st.scaler_state()
    return None


def load_loss_scaler_state(state):
    """restores a saved loss-scaler state; applied to the live step, or to the next one that is built"""
    global _PENDING_SCALER
    _PENDING_SCALER = dict(state) if state else None
    for st in _STEPS.values():
        _apply_scaler(st)


def _apply_scaler(st):
    global _PENDING_SCALER
    if _PENDING_SCALER is not None and st.scaler is not None:
        st._merge_skipped_updates()     # the skip counters in scaler[5:7] are overwritten below
        s = float(_PENDING_SCALER["scale"])
        st.scaler.copy_(torch.tensor([s, float(_PENDING_SCALER.get("growth_tracker", 0)), 0.0, 0.0, 1.0 / s, 0.0, 0.0, 0.0]))
        _PENDING_SCALER = None


def get_step(generator_F, discriminator_F, B, T, h, args, device, dtype_t=None, use_graph=None):
    dtype_t = dtype_t or compute_dtype(args)
    if use_graph is None:
        use_graph = tuning.current().graph
    Ge, De = generator_F.engine(dtype_t), discriminator_F.engine(dtype_t)
    if float(getattr(args, "vgg_scaling", -1.0)) > 0.0 and getattr(args, "tg_vgg", None) is None:
        from .models import VGG19          # the frozen feature extractor is built ONCE and kept on args
        args.tg_vgg = VGG19(args).to(device)
    key = (id(Ge), id(De), B, T, h, use_graph, bool(getattr(args, "pingpang", False)), id(getattr(args, "tg_fnet", None)),
           bool(getattr(args, "tg_fnet_train", False)),
           id(getattr(args, "tg_vgg", None)) if float(getattr(args, "vgg_scaling", -1.0)) > 0.0 else None)
    st = _STEPS.get(key)
    if st is None:
        for old in _STEPS.values():  # one live configuration: activation buffers are large - the old step's buffer sets are
            old.close()              # released BEFORE the new step allocates its own
        _STEPS.clear()
        pg, world = parallel.dist_info()
        st = TecoGANStep(Ge, De, B, T, h, args, device, use_graph=use_graph, process_group=pg, world_size=world)
        _STEPS[key] = st
        _apply_scaler(st)
    return st


def sync_optimizer_steps(optimizer_g, optimizer_d):
    """fp16 mode: Adam's `step` as torch counts it.  The host counts calls; an update skipped on overflow is counted on the
    device (tg_scaler_update) and subtracted inside tg_adam_scaled, because GradScaler.step() does not call
    optimizer.step() then (code/train.py:337,341).  Before optimizer.state_dict() is written the two are merged: the skip
    counts move from the device into the optimisers' step tensors.  Synchronises; no-op outside fp16 mode."""
    for st in _STEPS.values():
        st._opts = (optimizer_g, optimizer_d)
        st._merge_skipped_updates()


def TecoGAN(r_inputs, r_targets, discriminator_F, generator_F, args, Global_step, counter1, counter2, optimizer_g,
            optimizer_d, GAN_FLAG=True):
    """code/train.py:49-370."""
    if not GAN_FLAG:
        raise NotImplementedError("GAN_FLAG=False leaves discrim_loss undefined in the reference (code/train.py:340)")
    if not r_inputs.is_cuda:
        raise L.TecoganHipError("FRVSR_Train needs device tensors; there is no CPU path")
    B, T = r_inputs.shape[0], r_inputs.shape[1]
    if int(args.RNN_N) != T:
        raise ValueError("r_inputs.shape[1] must equal args.RNN_N")
    if T // 3 != 3 and not getattr(args, "pingpang", False) and not getattr(args, "tg_extend", False):
        raise RuntimeError("the reference's D-input reshape only works for RNN_N in {9,10,11} (code/train.py:143-145); "
                           "args.tg_extend=True opts into the documented extension (DESIGN.md, parity unpinned)")
    h = int(args.crop_size)
    st = get_step(generator_F, discriminator_F, B, T, h, args, r_inputs.device)
    _bind_optimizer(optimizer_g, generator_F)
    _bind_optimizer(optimizer_d, discriminator_F)
    st._opts = (optimizer_g, optimizer_d)   # (close() / a scaler reload merge the device-side skip counts into their step tensors)
    gg, gd = optimizer_g.param_groups[0], optimizer_d.param_groups[0]
    st.adam_t = [int(optimizer_g._tg_step), int(optimizer_d._tg_step), 0]
    f_hyper, opt_f = None, None
    if st.F_train:
        # the third optimiser of main.py:244-245 (commented out there): args.tg_fnet_optimizer, a torch.optim.Adam over
        # args.tg_fnet.parameters(); FRVSR_Train's signature (code/train.py:374-377) has no slot for it
        opt_f = getattr(args, "tg_fnet_optimizer", None)
        if opt_f is None:
            raise ValueError("args.tg_fnet_train needs args.tg_fnet_optimizer (torch.optim.Adam over args.tg_fnet.parameters())")
        _bind_optimizer(opt_f, args.tg_fnet)
        gf = opt_f.param_groups[0]
        f_hyper = (gf["lr"], gf["betas"], gf["eps"])
        st.adam_t[2] = int(opt_f._tg_step)
    st.run(r_inputs.float(), r_targets.float(), Global_step, gg["lr"], gd["lr"], gg["betas"], gd["betas"], gg["eps"],
           gd["eps"], f_hyper=f_hyper)
    optimizer_g._tg_step += 1
    optimizer_d._tg_step += 1
    if opt_f is not None:
        opt_f._tg_step += 1
    return _network(st, args, Global_step + 1, counter1, counter2)


def _network(st, args, global_step, counter1, counter2):
    """Network tuple of code/train.py:354-370 from the device scalars written by tg_loss_finalize (update_list, its EMA
    and tb are computed in that kernel; nothing here launches work besides one 192-byte copy, and nothing synchronises)."""
    s = st.out_scalars   # (a per-call copy made on lane B: step._emit_scalars)
    names = []
    if args.D_LAYERLOSS:
        names += ["D_layer_%d_loss" % i for i in range(4)] + ["D_layer_loss_sum"]
    names += ["l2_content_loss", "l2_warp_loss"]  # l2_content_loss holds the aliased total (code/train.py:244,293,299)
    if st.V is not None:  # code/train.py:271-273: one entry per tap layer, then their sum
        names += ["vgg_loss_2", "vgg_loss_3", "vgg_loss_4", "vgg_all"]
    if getattr(args, "pingpang", False):
        names += ["PingPang"]
    names += ["t_adversarial_loss", "t_discrim_loss", "t_discrim_real_output", "t_discrim_fake_output", "All_loss_Gen"]
    n = len(names)
    vals = [s[16 + i] for i in range(n)]
    avg = [s[40 + i] for i in range(n)]
    tb = s[14]
    avg += [tb, torch.tensor(st.dt_ratio), counter1, counter2]
    names_all = names + ["t_balance", "Dst_ratio", "withD_counter", "w_o_D_counter"]
    gen_loss = s[5]
    return Network(gen_output=st.out_gen, learning_rate=args.learning_rate, update_list=vals, update_list_name=names_all,
                   update_list_avg=avg, global_step=global_step, d_loss=s[8], gen_loss=gen_loss, fnet_loss=gen_loss,
                   tb=tb, target=st.target)


def FRVSR_Train(r_inputs, r_targets, args, discriminator_F, generator_F, step, counter1, counter2, optimizer_g,
                optimizer_d):
    """code/train.py:374-377."""
    return TecoGAN(r_inputs, r_targets, discriminator_F, generator_F, args, step, counter1, counter2, optimizer_g,
                   optimizer_d)
