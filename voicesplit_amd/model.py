"""``VoiceSplit`` / ``VoiceFilter`` behind the reference's own ``nn.Module`` surface.

Drop-in for models/voicesplit/model.py:9-89 and models/voicefilter/model.py:11-90:

* constructor ``Model(config)`` reading ``config.audio[config.audio['backend']]['num_freq']`` and
  ``config.model['emb_dim'|'lstm_dim'|'fc1_dim'|'fc2_dim']``;
* ``forward(x[B,T,num_freq], speaker_embedding[B,emb_dim]) -> mask[B,T,fc2_dim]``, called
  positionally by train.py:94 and utils/generic_utils.py:495,545;
* identical ``state_dict`` keys/shapes (``conv.{1,2,5,6,...,28,29}.*``, ``lstm.*``, ``fc1.*``,
  ``fc2.*``) so reference checkpoints load with ``strict=True`` (train.py:46) and
  ``set_init_dict`` (utils/generic_utils.py:647-679) matches by key and numel;
* ``.train()`` / ``.eval()`` switch BatchNorm between batch and running statistics.

The torch sub-modules below are *parameter containers only* -- they give the reference's default
initialisation (same RNG consumption order as the upstream constructor) and key names.  Their
``forward`` is never called: ``forward`` hands raw device pointers to ``vs_forward`` in
``libvoicesplit_hip.so``.  Inputs on a CPU device raise; there is no fallback.
"""
import weakref

import torch
import torch.nn as nn

from . import ops

# (cin, cout, (kt, kf), time dilation) -- models/voicesplit/model.py:15-52
_CONV_TABLE = (
    (1, 64, (1, 7), 1), (64, 64, (7, 1), 1), (64, 64, (5, 5), 1), (64, 64, (5, 5), 2),
    (64, 64, (5, 5), 4), (64, 64, (5, 5), 8), (64, 64, (5, 5), 16), (64, 8, (1, 1), 1),
)


class _Slot(nn.Identity):
    """Occupies the Sequential index of a ZeroPad2d / activation module of the reference."""


class _MaskForward(torch.autograd.Function):
    """Autograd node around the HIP path: ``forward`` = vs_forward_train (keeps the tape of saved
    activations), ``backward`` = vs_backward.  Replaces the graph autograd would record through
    models/voicesplit/model.py:66-89 when train.py:110 calls ``loss.backward()``."""

    @staticmethod
    def forward(ctx, module, x, dvec, *params):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError(
                "voicesplit_amd: the gradient wrt the input spectrogram is not produced (the reference "
                "never asks for it: x is data, train.py:85-94); detach x")
        x = x.detach().contiguous()
        dvec_c = dvec.detach().contiguous()
        dims = module._dims(x.shape[0], x.shape[1])
        sd = module._tensors()
        tape = ops.new_tape(dims, x.device)
        mask = ops.forward_train(sd, x, dvec_c, dims, module.conv_act, module.training, tape)
        module._bump_bn_counters()
        ctx.module, ctx.dims, ctx.tape = module, dims, tape
        # for lstm_status(): a WEAK reference -- a strong one would keep a 25-49 GB tape alive past its recycling (and past
        # the pool cap), beside the next, larger one when B or T grows, and inside every deepcopy / pickle of the module
        module.__dict__["_last_tape"] = (weakref.ref(tape), dims)
        ctx.training = module.training
        ctx.names = [n for n, _ in module._named_params()]
        ctx.save_for_backward(x, dvec_c, mask)
        return mask

    @staticmethod
    def backward(ctx, grad_mask):
        x, dvec, mask = ctx.saved_tensors
        module = ctx.module
        if ctx.tape is None:
            raise RuntimeError("voicesplit_amd: backward called twice on the same forward (the tape was released)")
        sd = module._tensors()
        sink = module.__dict__.get("_grad_sink")
        grads = ops.backward(sd, x, dvec, ctx.dims, module.conv_act, ctx.training, ctx.tape, mask,
                             grad_mask.contiguous(), want_dvec=ctx.needs_input_grad[2], sink=sink,
                             leaves_event=module.__dict__.get("_leaves_event"))
        ops.recycle_tape(ctx.tape)                           # 49 GB at B=64: back to the pool for the next forward
        ctx.tape = None
        out = [None, None, grads.get("speaker_embedding")]
        for i, name in enumerate(ctx.names):
            # a gradient the library wrote into the sink is already where .grad lives: nothing for autograd to accumulate
            out.append(grads[name] if ctx.needs_input_grad[3 + i] and not (sink and name in sink) else None)
        return tuple(out)


class _MaskNet(nn.Module):
    conv_act = "mish"

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.audio = self.config.audio[self.config.audio["backend"]]
        layers = []
        for i, (cin, cout, ks, dil) in enumerate(_CONV_TABLE):
            if i < 7:
                layers.append(_Slot())                       # ZeroPad2d slot
            layers.append(nn.Conv2d(cin, cout, kernel_size=ks, dilation=(dil, 1)))
            layers.append(nn.BatchNorm2d(cout))
            layers.append(_Slot())                           # Mish / ReLU slot
        self.conv = nn.Sequential(*layers)
        self.lstm = nn.LSTM(8 * self.audio["num_freq"] + self.config.model["emb_dim"],
                            self.config.model["lstm_dim"], batch_first=True, bidirectional=True)
        self.fc1 = nn.Linear(2 * self.config.model["lstm_dim"], self.config.model["fc1_dim"])
        self.fc2 = nn.Linear(self.config.model["fc1_dim"], self.config.model["fc2_dim"])

    # -- helpers ------------------------------------------------------------------------------
    def _dims(self, B, T):
        m = self.config.model
        return ops.make_dims(B, T, self.audio["num_freq"], m["emb_dim"], m["lstm_dim"], m["fc1_dim"], m["fc2_dim"])

    _TRANSIENT = ("_last_tape", "_grad_sink", "_leaves_event", "_tensor_index", "_tree_check", "_prepared")

    def __getstate__(self):
        # copy.deepcopy / torch.save(model): caches and views into other objects' memory (the last tape, a trainer's
        # gradient bucket, prepared eval weights) are not part of the module
        state = self.__dict__.copy()
        for k in self._TRANSIENT:
            state.pop(k, None)
        return state

    def __setattr__(self, name, value):
        # a replaced sub-module / parameter invalidates the cached tensor index (see _tensors)
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__.pop("_tensor_index", None)
        super().__setattr__(name, value)

    def _index_is_current(self) -> bool:
        """Is the module tree still the one the index was built from?  Identity of every child in its parent's
        ``_modules``, the NUMBER of children of every module (``model.fc1.add_module(...)`` adds one without touching an
        existing link) and identity + length of every module's ``_parameters`` / ``_buffers`` dict: ~35 modules, a few
        microseconds -- ``model.conv[i] = layer``, ``register_buffer`` / ``register_parameter`` on a child or a replaced
        ``_parameters`` dict all show up here (a replaced TENSOR needs no check: tensors are read through the dicts)."""
        links, dicts = self.__dict__["_tree_check"]
        for parent_modules, name, child in links:
            if parent_modules.get(name) is not child:
                return False
        for mod, pd, pl, bd, bl, ml in dicts:
            if mod._parameters is not pd or len(pd) != pl or mod._buffers is not bd or len(bd) != bl or len(mod._modules) != ml:
                return False
        return True

    def _tensors(self):
        """{state_dict key: tensor} of every parameter and persistent buffer.  The walk of named_parameters() /
        named_buffers() (~0.1 ms of host time in front of every step's first kernel) is done once: what is kept is (key,
        owning dict, name), and the tensors are read from the owning modules' dicts on every call -- load_state_dict,
        .to(), optimizer steps and parameter re-assignment are all seen; a change of the module TREE (new / replaced /
        removed sub-module, parameter or buffer anywhere below) is detected by ``_index_is_current`` and rebuilds it."""
        index = self.__dict__.get("_tensor_index")
        if index is None or not self._index_is_current():
            index, links, dicts = [], [], []
            for mod_name, mod in self.named_modules():
                prefix = mod_name + "." if mod_name else ""
                for n in mod._parameters:
                    if mod._parameters[n] is not None:
                        index.append((prefix + n, mod._parameters, n))
                for n in mod._buffers:
                    if mod._buffers[n] is not None and n not in mod._non_persistent_buffers_set:
                        index.append((prefix + n, mod._buffers, n))
                dicts.append((mod, mod._parameters, len(mod._parameters), mod._buffers, len(mod._buffers), len(mod._modules)))
                for cn, child in mod._modules.items():
                    links.append((mod._modules, cn, child))
            self.__dict__["_tensor_index"] = index
            self.__dict__["_tree_check"] = (links, dicts)
        return {k: d[n] for k, d, n in index}

    def set_gradient_sink(self, sink):
        """sink: {state_dict key: tensor} or None.  With a sink, ``loss.backward()`` OVERWRITES these tensors with the
        parameter gradients (vs_backward writes, it never accumulates) and hands autograd nothing for those parameters:
        when they are the ``.grad`` tensors themselves (views into the trainer's all-reduce bucket,
        sharding.GradientBucket) the 47 ``grad += new`` launches and the zeroing of a step disappear.  The caller owns
        the zero_grad semantics: every backward replaces the contents (no accumulation over several backward calls).
        The sink is hidden state: whoever installs it removes it (``Trainer.train_step`` installs it around its own
        ``loss.backward()`` only), otherwise a plain loop over the same model would see no ``.grad`` for these keys."""
        if sink is None:
            self.__dict__.pop("_grad_sink", None)
        else:
            self.__dict__["_grad_sink"] = dict(sink)

    def set_leaves_event(self, event):
        """event: a ``torch.cuda.Event`` (already recorded once, so that its handle exists) or None.  While set, every backward
        through this module records it when the gradients of the head and the BiLSTM are final (vs_grads.leaves_event): a
        data-parallel trainer starts the all-reduce of that part of its bucket there, beside the conv stack's backward.  Hidden
        state like the gradient sink: whoever installs it removes it."""
        if event is None:
            self.__dict__.pop("_leaves_event", None)
        else:
            handle = int(event.cuda_event)
            if not handle:
                raise ValueError("set_leaves_event: record the event once before installing it (its handle is created lazily)")
            self.__dict__["_leaves_event"] = handle

    def leaf_parameter_names(self):
        """state_dict keys of the parameters whose gradients are final at the leaves event, in named_parameters() order."""
        return [n for n, _ in self._named_params() if not n.startswith("conv.")]

    def _named_params(self):
        """[(key, parameter)] in named_parameters() order, from the same index (no module-tree walk)."""
        if "_tensor_index" not in self.__dict__ or not self._index_is_current():
            self._tensors()
        return [(k, d[n]) for k, d, n in self.__dict__["_tensor_index"] if isinstance(d[n], nn.Parameter)]

    def train(self, mode: bool = True):
        if mode:
            self.__dict__.pop("_prepared", None)     # the eval-mode weight cache is not kept through training
        return super().train(mode)

    def _run(self, x, dvec):
        x = x.contiguous()
        dvec = dvec.contiguous()
        dims = self._dims(x.shape[0], x.shape[1])
        sd = self._tensors()                      # only data pointers and version counters are read: no detach needed
        if not self.training:
            # eval mode (validation(), test.py, serving): the weight-only work of the forward is done once and
            # kept until a parameter or a running statistic changes (tensor identity + version counters)
            prep = self.__dict__.get("_prepared")
            if prep is None or not prep.matches(sd, dims):
                prep = ops.PreparedWeights(sd, dims)
                self.__dict__["_prepared"] = prep
            return ops.forward_prepared(sd, prep, x.detach(), dvec.detach(), dims, self.conv_act)
        self.__dict__.pop("_prepared", None)
        mask = ops.forward(sd, x.detach(), dvec.detach(), dims, self.conv_act, training=True)
        self._bump_bn_counters()
        return mask

    def _bump_bn_counters(self):
        if self.training:
            with torch.no_grad():
                bns = [m for m in self.conv if isinstance(m, nn.BatchNorm2d)]
                # ONE launch for the eight counters (eight `+= 1` were eight 6 us kernels between the head and the loss head: round 6,
                # tools/dispatch_census.py)
                torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
                for m in bns:
                    # running_mean/var were updated in place by the library through raw pointers, which torch cannot see: advance
                    # their version counters so that every cache keyed on them (ops.PreparedWeights) knows (no kernel launch)
                    torch.autograd.graph.increment_version(m.running_mean)
                    torch.autograd.graph.increment_version(m.running_var)

    def lstm_status(self):
        """0 when the persistent BiLSTM kernels of the last training forward / backward completed, 1 when one gave up
        (its output was NaN-poisoned; see vs_lstm_status), None when that is no longer knowable: no training forward has
        run, or its tape has been released since (the error word lives in the tape; the pool keeps ONE recycled tape alive,
        so the usual call -- right after a step -- finds it and reports the last launch on that buffer; with several forwards
        outstanding, or a graph freed without its backward, the buffer is gone).  Never reports success it cannot see.
        Synchronises."""
        last = self.__dict__.get("_last_tape")
        if last is None:
            return None
        ref, dims = last
        tape = ref()
        if tape is None:
            return None
        return ops.lstm_status(dims, tape=tape)

    def long_form_stages(self):
        """(conv_stage, sequence_stage) for ``streaming.separate_long_exact``: the conv stack on a batch
        of windows and BiLSTM + head on one full-length feature sequence, both in eval mode
        (running BatchNorm statistics), both straight into the C ABI stage entry points."""
        if self.training:
            raise RuntimeError("long-form inference runs in eval mode: call model.eval() first")
        sd = {k: v.detach() for k, v in self._tensors().items()}

        def conv_stage(xw):
            return ops.conv_stack(sd, xw.contiguous(), self._dims(xw.shape[0], xw.shape[1]), self.conv_act)

        def sequence_stage(feat, dvec):
            dims = self._dims(feat.shape[0], feat.shape[1])
            return ops.head(sd, ops.bilstm(sd, feat.contiguous(), dvec.contiguous(), dims), dims)

        return conv_stage, sequence_stage

    def forward(self, x, speaker_embedding):
        # x: [B, T, num_freq]; speaker_embedding: [B, emb_dim]  ->  mask [B, T, fc2_dim]
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError(
                "voicesplit_amd: the gradient wrt the input spectrogram is not produced (the reference "
                "never asks for it: x is data, train.py:85-94); detach x")
        if torch.is_grad_enabled():
            params = [p for _, p in self._named_params()]
            if speaker_embedding.requires_grad or any(p.requires_grad for p in params):
                return _MaskForward.apply(self, x, speaker_embedding, *params)
        return self._run(x, speaker_embedding)


class VoiceSplit(_MaskNet):
    """models/voicesplit/model.py:9 -- Mish in the conv stack (utils/generic_utils.py:376-399)."""
    conv_act = "mish"


class VoiceFilter(_MaskNet):
    """models/voicefilter/model.py:11 -- ReLU in the conv stack."""
    conv_act = "relu"
