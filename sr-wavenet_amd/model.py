"""Host-side mirror of the reference's model classes (``/root/reference/model.py``) on the MI355X engine.

Same constructor and method signatures, NumPy in / NumPy out like the reference's ``sess.run``
wrappers, but no TensorFlow: the graph body is the fixed kernel sequence of ``engine.WaveNetEngine``.

* ``WaveNet``            -- model.py:8-72 (clip-level softmax classifier), complete.
* ``WaveNetTeacher``     -- the 30-layer mu-law softmax teacher BASELINE.json names (decoder stack of
                            model.py:158-196 with the 256-way softmax head the reference carries at
                            model.py:100-112); this is the benchmark path.
* ``WaveNetAutoEncoder`` -- model.py:75-285: non-causal encoder + conditioned mixture-of-logistics decoder
                            (encoder.AutoEncoderEngine), the teacher teacher.py trains.
* ``ParallelWaveNet``    -- model.py:290-656: the IAF student distilled against that frozen teacher
                            (student.StudentEngine).
"""
from __future__ import annotations

import os
import time
from typing import Dict, Optional

import numpy as np
import torch

from . import kernels as K
from .engine import StackConfig, WaveNetEngine


def _train_step(eng):
    """One training step of an engine: the first two run as eager launches, then the step is captured as hipGraphs
    (inputs live in persistent device buffers, so replays see each new batch) -- at the reference scripts' small
    shapes the ~100-400 launches of a step cost more than the kernels.  SRWN_MODEL_GRAPHS=0 keeps eager launches."""
    if getattr(eng, "_graph_ready", False):
        return eng.train_step_graphed()
    out = eng.train_step()
    eng._eager_steps = getattr(eng, "_eager_steps", 0) + 1
    if eng._eager_steps >= 2 and os.environ.get("SRWN_MODEL_GRAPHS", "1") != "0" and not getattr(eng, "_graph_failed", False):
        try:
            eng.capture_graphs()
            eng._graph_ready = True
        except Exception as e:   # keep training with eager launches, say why once
            eng._graph_failed = True
            print("hipGraph capture failed (%s); continuing with eager launches" % e)
    return out


def _default_dtype():
    return torch.float32 if os.environ.get("SRWN_DTYPE", "bf16").lower() in ("f32", "fp32", "float32") else torch.bfloat16


# --- checkpoint files --------------------------------------------------------------------------------------------
# Two on-disk forms behind the reference's `checkpoint` state file (model.py:217-235): this package's own
# `model.ckpt-N.pt` (a torch state dict keyed by the reference's variable names) and TensorFlow's V2 bundle
# `model.ckpt-N.index` + `.data-00000-of-00001` as tf.train.Saver writes it (tf_checkpoint.py; SRWN_CKPT_FORMAT=tf or
# save(..., fmt="tf")), so weights trained with the reference load by name and vice versa.
def _write_state(logdir, global_step, params, fmt=None):
    fmt = fmt or os.environ.get("SRWN_CKPT_FORMAT", "pt")
    os.makedirs(logdir, exist_ok=True)
    if fmt == "tf":
        from . import tf_checkpoint as tfc
        name = "model.ckpt-%d" % int(global_step)
        tfc.write_bundle(os.path.join(logdir, name), {k: v.detach().float().cpu().numpy() for k, v in params.items()})
        tfc.write_checkpoint_state(logdir, name)
        return
    if fmt != "pt":
        raise ValueError("checkpoint format %r (pt, tf)" % (fmt,))
    state = {k: v.detach().cpu().clone() for k, v in params.items()}
    torch.save(state, os.path.join(logdir, "model.ckpt-%d.pt" % int(global_step)))
    with open(os.path.join(logdir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "model.ckpt-%d.pt"\n' % int(global_step))


def _read_state(logdir, params_fn):
    """None: no checkpoint state file; False: the file it names is missing; True: the tensors of `params_fn()` filled
    (it is only called once a file is there: building the name map instantiates the model's first engine)."""
    if logdir is None or not os.path.exists(os.path.join(logdir, "checkpoint")):
        return None
    from . import tf_checkpoint as tfc
    path = tfc.latest_checkpoint(logdir)
    if path is None or not (os.path.exists(path) or tfc.is_bundle(path)):
        print("Could not find checkpoint at %s" % path)
        return False
    params = params_fn()
    if tfc.is_bundle(path):
        arrays = tfc.read_bundle(path, names=list(params))       # Adam slots and counters in the file are ignored
        get = lambda k: torch.from_numpy(arrays[k])
    else:
        state = torch.load(path, weights_only=True)
        get = lambda k: state[k]
    for k, dst in params.items():
        src = get(k)
        if tuple(src.shape) != tuple(dst.shape):     # same element count in another layout would load scrambled weights
            raise ValueError("checkpoint variable %s has shape %s, the model's is %s" % (k, tuple(src.shape), tuple(dst.shape)))
        dst.copy_(src.to(dst.device))
    return True


class _EngineOwner:
    """Builds one engine per (batch, length) seen, all sharing the same parameters."""

    def _setup(self, cfg: StackConfig, seed: int):
        if not torch.cuda.is_available():
            raise RuntimeError("sr-wavenet_amd needs an MI355X (HIP) device; there is no CPU fallback")
        self._cfg = cfg
        self._seed = seed
        self._engines: Dict[tuple, WaveNetEngine] = {}
        self._primary: Optional[WaveNetEngine] = None
        self.last_checkpoint_time = time.time()

    def _engine(self, B: int, T: int) -> WaveNetEngine:
        key = (int(B), int(T))
        eng = self._engines.get(key)
        if eng is None:
            eng = WaveNetEngine(self._cfg, B, T, "cuda", seed=self._seed, share_from=self._primary)
            if self._primary is None:
                self._primary = eng
            self._engines[key] = eng
        return eng

    @property
    def network_params(self):
        """Reference name -> tensor (the analogue of tf.get_collection(TRAINABLE_VARIABLES, scope))."""
        if self._primary is None:
            self._engine(1, self._default_length)
        return self._primary.tf_variables(self._scope, decoder=self._decoder_names)

    # --- checkpointing with the reference's cadence semantics (model.py:217-239) -------------------
    def save(self, logdir, global_step, force=False, fmt=None):
        if force or time.time() - self.last_checkpoint_time > 60:
            _write_state(logdir, global_step, self.network_params, fmt)
            self.last_checkpoint_time = time.time()
            return True
        return False

    def load(self, logdir):
        ok = _read_state(logdir, lambda: self.network_params)
        if ok:
            self._primary.repack()
            print("Restoring previous session")
        return ok


class WaveNet(_EngineOwner):
    """model.py:8-72.  ``train(inputs[B,T], targets[B,output_size]) -> loss``; ``predict -> [B,1,C]``."""

    def __init__(self, input_size, output_size, dilations, filter_width=2, dilation_channels=32, skip_channels=256,
                 output_channels=256, name="WaveNet", learning_rate=0.001, dtype=None, seed=0):
        self.input_size = input_size
        self.output_size = output_size
        self.dilations = dilations
        self.filter_width = filter_width
        self.dilation_channels = dilation_channels
        self.skip_channels = skip_channels
        self.output_channels = output_channels
        if output_size != output_channels:
            # the reference's softmax CE (model.py:29) needs logits and labels of equal width
            raise ValueError("output_size (%d) must equal output_channels (%d)" % (output_size, output_channels))
        self._scope, self._decoder_names, self._default_length = name, False, int(input_size)
        self._setup(StackConfig(dilations=list(dilations), filter_width=filter_width,
                                dilation_channels=dilation_channels, skip_channels=skip_channels,
                                output_channels=output_channels, shift_input=False, head_mode="pooled",
                                dtype=dtype or _default_dtype(), learning_rate=learning_rate), seed)

    def _stage(self, inputs, targets=None):
        x = np.asarray(inputs, dtype=np.float32)
        if x.ndim != 2:
            raise ValueError("inputs must be [batch, samples]")
        if x.shape[1] != self.input_size:
            # tf.nn.pool window = input_size, VALID (model.py:58): other lengths would pool differently
            raise ValueError("inputs have %d samples, the model was built for input_size=%d" % (x.shape[1], self.input_size))
        eng = self._engine(x.shape[0], x.shape[1])
        t = None
        if targets is not None:
            t = torch.as_tensor(np.asarray(targets, dtype=np.float32), device="cuda")
            if tuple(t.shape) != (x.shape[0], self.output_size):
                raise ValueError("targets must be [batch, %d]" % self.output_size)
        eng.set_inputs(torch.as_tensor(x, device="cuda"), t)
        return eng

    def train(self, inputs, targets):
        eng = self._stage(inputs, targets)
        _train_step(eng)
        return np.float32(eng.loss.item())

    def predict(self, inputs):
        eng = self._stage(inputs)
        eng.forward(with_loss=False)
        return eng.probs.cpu().numpy()[:, None, :]


class WaveNetTeacher(_EngineOwner):
    """The mu-law softmax teacher of BASELINE.json configs[1-2]: ``createDecoder``'s stack
    (model.py:158-196: RightShift teacher forcing, per-layer conditioning add) with a
    ``quantization_channels``-way softmax over mu-law codes per sample (model.py:100-112).

    ``train(inputs[B,T], encoding=None, conditions=None) -> loss`` (mean CE over B*T);
    ``get_logits`` / ``predict_codes`` for evaluation.  ``encoding`` is [B, T/pool_stride, latent];
    ``conditions`` [B, condition_size] is tiled over frames and concatenated (model.py:161-167).
    """

    def __init__(self, input_size, condition_size, dilations, filter_width=2, dilation_channels=32,
                 skip_channels=256, quantization_channels=256, latent_channels=16, pool_stride=512,
                 name="WaveNetTeacher", learning_rate=0.001, use_encoding=False, dtype=None, seed=0,
                 head="softmax", num_mixtures=5):
        self._ctor = dict(input_size=int(input_size), condition_size=int(condition_size),
                          dilations=[int(d) for d in dilations], filter_width=int(filter_width),
                          dilation_channels=int(dilation_channels), skip_channels=int(skip_channels),
                          quantization_channels=int(quantization_channels), latent_channels=int(latent_channels),
                          pool_stride=int(pool_stride), name=name, learning_rate=float(learning_rate),
                          use_encoding=bool(use_encoding), seed=int(seed), head=head, num_mixtures=int(num_mixtures))
        self.input_size = input_size
        self.condition_size = condition_size
        self.dilations = dilations
        self.quantization_channels = quantization_channels
        self.latent_channels = latent_channels
        self.pool_stride = pool_stride
        self.use_encoding = bool(use_encoding)
        self.head = head
        self.num_mixtures = num_mixtures
        if head not in ("softmax", "mol"):
            raise ValueError("head must be 'softmax' (mu-law classes) or 'mol' (mixture of logistics, model.py:114)")
        cond_ch = (latent_channels + condition_size) if self.use_encoding else 0
        self._scope, self._decoder_names, self._default_length = name, bool(cond_ch), int(input_size)
        self._setup(StackConfig(dilations=list(dilations), filter_width=filter_width,
                                dilation_channels=dilation_channels, skip_channels=skip_channels,
                                output_channels=quantization_channels if head == "softmax" else 4 * num_mixtures,
                                cond_channels=cond_ch, pool_stride=pool_stride if cond_ch else 1, shift_input=True,
                                head_mode="per_timestep" if head == "softmax" else "mol",
                                dtype=dtype or _default_dtype(),
                                learning_rate=learning_rate), seed)

    def save(self, logdir, global_step, force=False, fmt=None):
        """Checkpoint + ``config.json`` (the constructor arguments; the reference gets them from the meta graph
        it imports at model.py:318)."""
        done = super().save(logdir, global_step, force, fmt)
        if done:
            import json
            with open(os.path.join(logdir, "config.json"), "w") as f:
                json.dump(self._ctor, f)
        return done

    @classmethod
    def from_checkpoint(cls, logdir, dtype=None):
        import json
        path = os.path.join(logdir, "config.json")
        if not os.path.exists(path):
            raise FileNotFoundError("%s: no config.json (save the teacher with WaveNetTeacher.save)" % logdir)
        m = cls(dtype=dtype, **json.load(open(path)))
        if not m.load(logdir):
            raise FileNotFoundError("%s: no checkpoint to restore" % logdir)
        return m

    def _stage(self, inputs, encoding=None, conditions=None):
        x = torch.as_tensor(np.asarray(inputs, dtype=np.float32), device="cuda")
        B, T = x.shape
        eng = self._engine(B, T)
        cond = None
        if self.use_encoding:
            if encoding is None:
                raise ValueError("this teacher was built with use_encoding=True; pass encoding [B, T/pool, latent]")
            e = torch.as_tensor(np.asarray(encoding, dtype=np.float32), device="cuda")
            if self.condition_size > 0:
                c = torch.as_tensor(np.asarray(conditions, dtype=np.float32), device="cuda")
                e = torch.cat([e, c[:, None, :].expand(-1, e.shape[1], -1)], dim=2)   # model.py:162-165
            cond = e.contiguous()
        codes = None
        if self.head == "softmax":
            codes = K.mu_law_encode(x.contiguous(), self.quantization_channels)        # ops.py:82-93
        eng.set_inputs(x, codes, cond)
        return eng

    def train(self, inputs, encoding=None, conditions=None):
        eng = self._stage(inputs, encoding, conditions)
        _train_step(eng)
        return np.float32(eng.loss.item())

    def get_logits(self, inputs, encoding=None, conditions=None):
        eng = self._stage(inputs, encoding, conditions)
        return eng.forward(want_logits=True).cpu().numpy()

    def loss(self, inputs, encoding=None, conditions=None):
        eng = self._stage(inputs, encoding, conditions)
        eng.forward()
        return np.float32(eng.loss.item())

    def generate(self, batch_size, num_samples, mode="sample", seed=0, forced=None, return_logits=False,
                 encoding=None, conditions=None):
        """Queue-cached autoregressive generation (the O(T L) replacement of the reference's O(T^2 L)
        loop, teacher.py:140-171): returns audio [B, num_samples] float32, or (audio, codes, logits) when
        return_logits.  `forced` [B, num_samples] = teacher forcing.  The softmax teacher emits mu-law decoded
        samples; the mixture-of-logistics teacher (optionally conditioned on `encoding`) emits logistic samples."""
        if self.head == "softmax" and self.use_encoding:
            raise NotImplementedError("generation: the conditioned softmax teacher is not built")
        eng = self._primary or self._engine(1, self._default_length)
        f = None if forced is None else torch.as_tensor(np.asarray(forced, dtype=np.float32), device="cuda")
        cond = None
        if self.use_encoding:
            if encoding is None:
                raise ValueError("this teacher was built with use_encoding=True; pass encoding [B, frames, latent]")
            cond = torch.as_tensor(np.asarray(encoding, dtype=np.float32), device="cuda")
            if self.condition_size > 0:
                c = torch.as_tensor(np.asarray(conditions, dtype=np.float32), device="cuda")
                cond = torch.cat([cond, c[:, None, :].expand(-1, cond.shape[1], -1)], dim=2)
            cond = cond.contiguous()
        a, c, lg = eng.generate(int(num_samples), mode=mode, seed=seed, forced=f, want_logits=return_logits,
                                batch=int(batch_size), cond=cond)
        if return_logits:
            return a.cpu().numpy(), c.cpu().numpy(), lg.cpu().numpy()
        return a.cpu().numpy()


class WaveNetAutoEncoder(object):
    """model.py:75-285 on ``encoder.AutoEncoderEngine``: the non-causal encoder (ResidualDilationLayerNC chain,
    skip sum -> latent 1x1 -> average pool) and the conditioned mixture-of-logistics decoder, trained jointly on
    ``discretized_mix_logistic_loss(inputs, logits)`` (model.py:103,114,116).

    NumPy in / NumPy out like the reference's ``sess.run`` wrappers; there is no session.  One (batch, length) per
    model object.  ``reconstruct*`` draw the sampler's uniforms on the device (``seed`` makes them repeatable)."""

    def __init__(self, input_size, condition_size, num_mixtures, dilations, filter_width=2, encoder_channels=128,
                 dilation_channels=32, skip_channels=256, latent_channels=16, pool_stride=512,
                 name="WaveNetAutoEncoder", learning_rate=0.001, dtype=None, seed=0):
        if not torch.cuda.is_available():
            raise RuntimeError("sr-wavenet_amd needs an MI355X (HIP) device; there is no CPU fallback")
        self._ctor = dict(input_size=int(input_size), condition_size=int(condition_size),
                          num_mixtures=int(num_mixtures), dilations=[int(d) for d in dilations],
                          filter_width=int(filter_width), encoder_channels=int(encoder_channels),
                          dilation_channels=int(dilation_channels), skip_channels=int(skip_channels),
                          latent_channels=int(latent_channels), pool_stride=int(pool_stride), name=name,
                          learning_rate=float(learning_rate), seed=int(seed))
        self.input_size = input_size
        self.condition_size = condition_size
        self.num_mixtures = num_mixtures
        self.dilations = dilations
        self.filter_width = filter_width
        self.encoder_channels = encoder_channels
        self.dilation_channels = dilation_channels
        self.skip_channels = skip_channels
        self.latent_channels = latent_channels
        self.pool_stride = pool_stride
        self._name, self._seed = name, seed
        self._cfg = StackConfig(dilations=list(dilations), filter_width=filter_width,
                                dilation_channels=dilation_channels, skip_channels=skip_channels,
                                output_channels=4 * num_mixtures, cond_channels=latent_channels + condition_size,
                                pool_stride=pool_stride, shift_input=True, head_mode="mol",
                                dtype=dtype or _default_dtype(), learning_rate=learning_rate)
        self._eng = None
        self._gen = None
        self.last_checkpoint_time = time.time()

    # ------------------------------------------------------------------------------------------------
    def _engine(self, B: int, T: int):
        from .encoder import AutoEncoderEngine
        if self._eng is None:
            self._eng = AutoEncoderEngine(self._cfg, B, T, self.encoder_channels, self.latent_channels,
                                          self.condition_size, "cuda", seed=self._seed)
        elif (self._eng.B, self._eng.T) != (int(B), int(T)):
            raise NotImplementedError("WaveNetAutoEncoder: one (batch, length) per model object for now; built for "
                                      "%s, got %s" % ((self._eng.B, self._eng.T), (B, T)))
        return self._eng

    def _stage(self, inputs, conditions):
        x = torch.as_tensor(np.asarray(inputs, dtype=np.float32), device="cuda")
        if x.ndim != 2:
            raise ValueError("inputs must be [batch, samples]")
        eng = self._engine(*x.shape)
        c = None
        if self.condition_size > 0:
            if conditions is None:
                raise ValueError("this auto-encoder was built with condition_size > 0; pass conditions")
            c = torch.as_tensor(np.asarray(conditions, dtype=np.float32), device="cuda")
        eng.set_inputs(x, c)
        return eng

    def _put_encoding(self, eng, encoding):
        e = torch.as_tensor(np.asarray(encoding, dtype=np.float32), device="cuda")
        d = eng.dec
        if tuple(e.shape) != (eng.B, d.frames, self.latent_channels):
            raise ValueError("encoding must be [batch, samples/pool_stride, latent_channels]")
        d.cond_in.view(eng.B, d.frames, d.Ep)[:, :, :self.latent_channels].copy_(e)

    def _sample(self, eng, seed=None):
        """``sample_from_discretized_mix_logistic`` on the decoder's logits (model.py:198; ops.py:178-201)."""
        d = eng.dec
        M = self.num_mixtures
        if self._gen is None:
            self._gen = torch.Generator(device="cuda")
            self._gen.manual_seed(self._seed)
        if seed is not None:
            self._gen.manual_seed(int(seed))
        u1 = torch.empty((d.N, M), dtype=torch.float32, device="cuda").uniform_(1e-5, 1.0 - 1e-5, generator=self._gen)
        u2 = torch.empty((d.N,), dtype=torch.float32, device="cuda").uniform_(1e-5, 1.0 - 1e-5, generator=self._gen)
        out = torch.empty((d.N,), dtype=torch.float32, device="cuda")
        from ._lib import call
        call("srwn_mol_sample", d.logits32.data_ptr(), d.logits32.stride(0), M, u1.data_ptr(), u2.data_ptr(),
             out.data_ptr(), d.N, torch.cuda.current_stream().cuda_stream)
        return out.view(eng.B, eng.T).cpu().numpy()

    @property
    def network_params(self):
        eng = self._eng or self._engine(1, self.input_size)
        out = dict(eng.enc.tf_variables(self._name + "/Encoder"))
        out.update(eng.dec.tf_variables(self._name + "/Decoder", decoder=True))
        return out

    # --- the reference's methods (model.py:217-285) ---------------------------------------------------
    def save(self, logdir, global_step, force=False, fmt=None):
        if force or time.time() - self.last_checkpoint_time > 60:
            import json
            _write_state(logdir, global_step, self.network_params, fmt)
            with open(os.path.join(logdir, "config.json"), "w") as f:
                json.dump(dict(self._ctor, **{"class": "WaveNetAutoEncoder"}), f)
            self.last_checkpoint_time = time.time()
            return True
        return False

    def load(self, logdir):
        ok = _read_state(logdir, lambda: self.network_params)
        if ok:
            self._eng.enc.repack(); self._eng.dec.repack()
            print("Restoring previous session")
        return ok

    @classmethod
    def from_checkpoint(cls, logdir, batch, length, dtype=None):
        import json
        cfg = json.load(open(os.path.join(logdir, "config.json")))
        cfg.pop("class", None)
        m = cls(dtype=dtype, **cfg)
        m._engine(batch, length)
        if not m.load(logdir):
            raise FileNotFoundError("%s: no checkpoint to restore" % logdir)
        return m

    def train(self, inputs, conditions=None):
        eng = self._stage(inputs, conditions)
        _train_step(eng)
        return np.float32(eng.loss.item())

    def encode(self, inputs, conditions=None):
        eng = self._stage(inputs, conditions)
        return eng.encode().view(eng.B, -1, self.latent_channels).cpu().numpy()

    def reconstruct(self, inputs, conditions=None, seed=None):
        """``self.out`` (model.py:268-273): encode, run the decoder teacher-forced on the same clip, sample."""
        eng = self._stage(inputs, conditions)
        eng.forward()
        return self._sample(eng, seed)

    def reconstruct_with_encoding(self, inputs, encoding, conditions=None, seed=None):
        eng = self._stage(inputs, conditions)
        self._put_encoding(eng, encoding)
        eng.dec.forward(with_loss=False)
        return self._sample(eng, seed)

    def get_logits(self, inputs, encoding, conditions=None):
        eng = self._stage(inputs, conditions)
        self._put_encoding(eng, encoding)
        return eng.dec.forward(want_logits=True, with_loss=False).cpu().numpy()

    def generate(self, encoding, conditions=None, num_samples=None, mode="sample", seed=0):
        """Queue-cached autoregressive sampling from the decoder given an encoding (the O(T L) replacement of the
        reference's sample-by-sample loop over ``reconstruct_with_encoding``, generator.py:150-170 /
        teacher.py:140-171): audio [B, num_samples] in [-1, 1]."""
        e = torch.as_tensor(np.asarray(encoding, dtype=np.float32), device="cuda")
        if e.ndim != 3 or e.shape[2] != self.latent_channels:
            raise ValueError("encoding must be [batch, frames, latent_channels]")
        B, frames = int(e.shape[0]), int(e.shape[1])
        T = int(num_samples) if num_samples is not None else frames * self.pool_stride
        if T > frames * self.pool_stride:
            raise ValueError("num_samples %d exceeds frames * pool_stride = %d" % (T, frames * self.pool_stride))
        if self.condition_size > 0:
            if conditions is None:
                raise ValueError("this auto-encoder was built with condition_size > 0; pass conditions")
            c = torch.as_tensor(np.asarray(conditions, dtype=np.float32), device="cuda")
            e = torch.cat([e, c[:, None, :].expand(-1, frames, -1)], dim=2)               # model.py:161-167
        eng = self._eng or self._engine(B, frames * self.pool_stride)
        a, _, _ = eng.dec.generate(T, mode=mode, seed=seed, batch=B, cond=e.contiguous())
        return a.cpu().numpy()

    def mu_law(self, inputs, conditions=None):
        raise AttributeError("WaveNetAutoEncoder.mu_law reads self.targets, which the reference never defines "
                             "(model.py:100,276): it raises there too")


class ParallelWaveNet(object):
    """model.py:290-656 on ``student.StudentEngine``: ``num_flows`` inverse-autoregressive flows distilled against a
    frozen mixture-of-logistics teacher.

    ``teacher`` is a ``WaveNetAutoEncoder`` (or a decoder-only ``WaveNetTeacher(head="mol", use_encoding=True)``), or
    the directory one was saved to (the reference takes the checkpoint directory and imports its meta graph,
    model.py:313-324).  The ``sess`` argument of
    every method is accepted for call compatibility with student.py and ignored (there is no session).
    ``encode`` / ``reconstruct`` run the teacher auto-encoder (model.py:644-656); ``train`` is the per-row-clipped
    slow path (model.py:599-632), ``train_fast`` the one student.py:107 uses."""

    def __init__(self, input_size, condition_size, dilations, teacher, num_flows=2, filter_width=2,
                 dilation_channels=32, skip_channels=256, latent_channels=16, pool_stride=512,
                 name="ParallelWaveNet", alpha=1.0, beta=1.0, gamma=1.0, learning_rate=0.001, dtype=None, seed=0):
        if not torch.cuda.is_available():
            raise RuntimeError("sr-wavenet_amd needs an MI355X (HIP) device; there is no CPU fallback")
        self.input_size = input_size
        self.condition_size = condition_size
        self.dilations = dilations
        self.teacher = teacher
        self.num_flows = num_flows
        self.filter_width = filter_width
        self.dilation_channels = dilation_channels
        self.skip_channels = skip_channels
        self.latent_channels = latent_channels
        self.pool_stride = pool_stride
        self._name = name
        self._abg = (float(alpha), float(beta), float(gamma))
        self._lr, self._seed = learning_rate, seed
        self._teacher_dir = None
        self._dtype = dtype
        if isinstance(teacher, (str, os.PathLike)):
            self._teacher_dir = os.fspath(teacher)
            import json
            cfgp = os.path.join(self._teacher_dir, "config.json")
            if not os.path.exists(cfgp):
                raise FileNotFoundError("%s: no config.json (save the teacher with its save() method)" % self._teacher_dir)
            if json.load(open(cfgp)).get("class") == "WaveNetAutoEncoder":
                self._teacher = None      # built on first use: the auto-encoder is tied to one (batch, length)
                self._teacher_cfg = {k: v for k, v in json.load(open(cfgp)).items() if k != "class"}
            else:
                self._teacher = WaveNetTeacher.from_checkpoint(self._teacher_dir, dtype=dtype)
        else:
            self._teacher = teacher
        t = self._teacher
        if t is None:
            tc = self._teacher_cfg
            tl, tcs, tp, tdt = tc["latent_channels"], tc["condition_size"], tc["pool_stride"], None
        elif isinstance(t, WaveNetAutoEncoder):
            tl, tcs, tp, tdt = t.latent_channels, t.condition_size, t.pool_stride, t._cfg.dtype
        elif isinstance(t, WaveNetTeacher) and t.head == "mol" and t.use_encoding:
            tl, tcs, tp, tdt = t.latent_channels, t.condition_size, t.pool_stride, t._cfg.dtype
        else:
            raise ValueError("teacher must be a WaveNetAutoEncoder, or a mixture-of-logistics WaveNetTeacher built "
                             "with use_encoding=True, or a directory one of them was saved to")
        if (tl, tcs, tp) != (latent_channels, condition_size, pool_stride):
            raise ValueError("student and teacher must agree on latent_channels, condition_size and pool_stride "
                             "(they share the encoding placeholders, model.py:318-324)")
        self._flow_cfg = StackConfig(dilations=list(dilations), filter_width=filter_width,
                                     dilation_channels=dilation_channels, skip_channels=skip_channels,
                                     cond_channels=latent_channels + condition_size, pool_stride=pool_stride,
                                     dtype=dtype or tdt or _default_dtype(), learning_rate=learning_rate)
        self._engines: Dict[tuple, object] = {}
        self._primary = None
        self.last_checkpoint_time = time.time()

    # ------------------------------------------------------------------------------------------------
    def _engine(self, B: int, T: int):
        from .student import StudentEngine
        key = (int(B), int(T))
        eng = self._engines.get(key)
        if eng is None:
            if self._primary is not None:
                raise NotImplementedError("ParallelWaveNet: one (batch, length) per model object for now; got %s "
                                          "after %s" % (key, next(iter(self._engines))))
            a, b, g = self._abg
            if self._teacher is None:
                self._teacher = WaveNetAutoEncoder.from_checkpoint(self._teacher_dir, B, T, dtype=self._dtype)
            teng = self._teacher._engine(B, T)
            eng = StudentEngine(teng.dec if isinstance(self._teacher, WaveNetAutoEncoder) else teng, self._flow_cfg,
                                self.num_flows, alpha=a, beta=b, gamma=g, learning_rate=self._lr, seed=self._seed)
            self._primary = eng
            self._engines[key] = eng
        return eng

    def _stage(self, inputs, truth, encoding, conditions):
        z = torch.as_tensor(np.asarray(inputs, dtype=np.float32), device="cuda")
        B, T = z.shape
        eng = self._engine(B, T)
        e = torch.as_tensor(np.asarray(encoding, dtype=np.float32), device="cuda")
        if self.condition_size > 0:
            if conditions is None:
                raise ValueError("this student was built with condition_size > 0; pass conditions [B, condition_size]")
            c = torch.as_tensor(np.asarray(conditions, dtype=np.float32), device="cuda")
            e = torch.cat([e, c[:, None, :].expand(-1, e.shape[1], -1)], dim=2)          # model.py:496-499
        if tuple(e.shape) != (B, T // self.pool_stride, self.latent_channels + self.condition_size):
            raise ValueError("encoding must be [batch, samples/pool_stride, latent_channels]")
        tr = None if truth is None else torch.as_tensor(np.asarray(truth, dtype=np.float32), device="cuda")
        eng.set_inputs(z, tr, e.contiguous())
        return eng

    @property
    def network_params(self):
        if self._primary is None:
            self._engine(1, self.input_size)
        out = {}
        for i, f in enumerate(self._primary.flows):
            out.update(f.tf_variables("%s/Flow%d/Flow%d" % (self._name, i, i)))           # model.py:417,468,510
        return out

    # --- checkpointing (model.py:540-567) ---------------------------------------------------------------
    def load(self, sess, logdir):
        if self._teacher_dir is not None and self._teacher is not None:
            self._teacher.load(self._teacher_dir)                                         # model.py:543-544
        ok = _read_state(logdir, lambda: self.network_params)
        if ok:
            for f in self._primary.flows:
                f.repack()
            print("Restoring previous session")
        return ok

    def save(self, sess, logdir, global_step, force=False, fmt=None):
        if force or time.time() - self.last_checkpoint_time > 60:
            _write_state(logdir, global_step, self.network_params, fmt)
            self.last_checkpoint_time = time.time()
            return True
        return False

    # --- graph outputs (model.py:570-597) ---------------------------------------------------------------
    def generate(self, sess, inputs, encoding, conditions=None):
        """noise [B,T] -> audio [B,T,1] in [-1,1] in ONE parallel pass (``self.out``, model.py:535)."""
        eng = self._stage(inputs, None, encoding, conditions)
        eng.forward_flows()
        return eng.out.view(eng.B, eng.T, 1).cpu().numpy()

    def getEntropy_fast(self, sess, inputs, encoding, conditions=None):
        """sum(log s_tot + 2) over the batch (model.py:356)."""
        eng = self._stage(inputs, None, encoding, conditions)
        eng.forward_flows()
        return np.float32(float(eng.logs.item()) + 2.0 * eng.N)

    def getEntropy(self, sess, inputs, encoding, conditions=None):
        """Per-sample entropies [B].  (The reference feeds one noise row against the whole batch of encodings,
        model.py:584-590; here every row is paired with its own encoding.)"""
        eng = self._stage(inputs, None, encoding, conditions)
        eng.forward_flows()
        logs = sum(f.prm[:, 0].view(eng.B, eng.T).sum(1) for f in eng.flows)
        return (logs + 2.0 * eng.T).double().cpu().numpy()

    def train_fast(self, sess, inputs, truth, encoding, conditions=None):
        """One distillation step (model.py:634-642): returns (loss, power_loss)."""
        eng = self._stage(inputs, truth, encoding, conditions)
        _train_step(eng)
        l = eng.losses()
        return np.float32(l["loss"]), np.float32(l["power_loss"])

    def train(self, sess, inputs, truth, encoding, conditions=None):
        """The slow path (model.py:599-632): per-noise-row gradients, each clipped to norm 1, then averaged and
        applied; returns (mean loss, mean power loss).  student.py:107 trains with ``train_fast``."""
        eng = self._stage(inputs, truth, encoding, conditions)
        l, p = eng.train_per_sample()
        return np.float32(l), np.float32(p)

    def _ae_teacher(self, inputs):
        if self._teacher is None:
            x = np.asarray(inputs)
            self._teacher = WaveNetAutoEncoder.from_checkpoint(self._teacher_dir, x.shape[0], x.shape[1], dtype=self._dtype)
        if not isinstance(self._teacher, WaveNetAutoEncoder):
            raise NotImplementedError("encode/reconstruct run the teacher's encoder (model.py:644-656): build the "
                                      "student on a WaveNetAutoEncoder teacher")
        return self._teacher

    def encode(self, sess, inputs, conditions=None):
        """The teacher's encoding of a clip (``teacher_encoding``, model.py:644-649)."""
        return self._ae_teacher(inputs).encode(inputs, conditions)

    def reconstruct(self, sess, inputs, conditions=None):
        """The teacher's own reconstruction (``teacher_out``, model.py:651-656)."""
        return self._ae_teacher(inputs).reconstruct(inputs, conditions)
