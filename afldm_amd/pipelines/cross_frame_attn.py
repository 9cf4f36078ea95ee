"""Cross-frame ("equivariant") attention hook — surface of reference
afldm/pipelines/cross_frame_attn.py (AttnState :6-51, CrossFrameAttnProcessor :54-130,
get_/set_unet_attn_processor :133-190) over afldm_amd's Attention modules.

STORE: remember the pre-norm input of every self-attention, keyed by timestep; LOAD: take K and
V from the remembered (unshifted-pass) map, group-normed with the layer's own GroupNorm, while
Q comes from the current hidden states.  Tensors are NHWC inside the UNet forward."""
import torch

from .. import ops
from ..models.blocks import AttnProcessor2_0, packed_norm


class AttnState:
    """What the cross-frame processors of one UNet share (API of the reference's AttnState, cross_frame_attn.py:6-51):
    the mode - STORE the attention inputs of this pass / LOAD the stored ones as K, V / IDLE - plus the timestep that keys
    the stored maps, the slot a STORE pass writes (`store_id`, two slots for the interpolating processor) and the blend
    weight `alpha` of a LOAD pass.  `state`, `timestep`, `store_id` and `alpha` are read-only views; they change through
    the set_* / to_* methods only, exactly as in the reference."""
    STORE, LOAD, IDLE = 0, 1, 2
    _FIELDS = ("state", "timestep", "store_id", "alpha")

    def __init__(self):
        self._v = {}
        self.reset()

    def reset(self):
        self._v.update(state=self.STORE, timestep=0, store_id=0, alpha=0)

    def __getattr__(self, name):                      # only reached for names that are not real attributes
        if name in AttnState._FIELDS:
            return self.__dict__["_v"][name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in AttnState._FIELDS:
            raise AttributeError(f"AttnState.{name} is read-only: use set_{name}() / to_load() / to_idle() / reset()")
        object.__setattr__(self, name, value)

    def set_timestep(self, t):
        self._v["timestep"] = t.item() if torch.is_tensor(t) else t

    def set_alpha(self, alpha):
        self._v["alpha"] = alpha

    def set_store_id(self, store_id):
        self._v["store_id"] = store_id

    def to_load(self):
        self._v["state"] = self.LOAD

    def to_idle(self):
        self._v["state"] = self.IDLE


class CrossFrameAttnProcessor(AttnProcessor2_0):
    """cache_kv (not in the reference; the graph-replayed harness sets it): a STORE pass also keeps the keys / values it
    projects from its own hidden states, and a LOAD pass attends to them directly instead of group-norming and projecting the
    stored map again every step (reference cross_frame_attn.py:88-125 recomputes them: same layer, same input, same numbers).
    Without interpolation only; `maps` is still filled as the reference does."""

    def __init__(self, attn_state: AttnState, enable_interp=False, cache_kv=False):
        super().__init__()
        self.attn_state = attn_state
        self.maps = [dict(), dict()]
        self.kv = [dict(), dict()]
        self.enable_interp = enable_interp
        self.cache_kv = bool(cache_kv) and not enable_interp

    def _kv_source(self, attn, stored, batch):
        """group-normed [Bk, HW, C] tokens of a stored NHWC map; Bk must divide the batch (the
        kernel indexes kv-sample b // (B/Bk): the reference's batch repeat, cross_frame_attn.py:91-96)."""
        Bk, H, W, C = stored.shape
        if batch % Bk != 0:
            raise ValueError(f"stored map batch {Bk} does not divide the current batch {batch}")
        gn = attn.group_norm
        if gn is None:
            return stored.view(Bk, H * W, C)
        gamma, beta = packed_norm(gn)
        stats = ops.gn_stats(stored, gn.num_groups)
        return ops.gn_apply(stored, stats, gamma, beta, gn.num_groups, gn.eps, act=0).view(Bk, H * W, C)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        if encoder_hidden_states is not None:       # not self-attention: vanilla
            return super().__call__(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        st, t = self.attn_state.state, self.attn_state.timestep
        if st == AttnState.IDLE:
            return super().__call__(attn, hidden_states, None, attention_mask, temb)
        if st == AttnState.STORE:
            sid = self.attn_state.store_id
            # (under graph capture the tensor lives in the graph's pool for as long as this dict holds it and nobody writes it
            #  in place: no copy launch; eager passes keep the reference's clone)
            capturing = torch.cuda.is_current_stream_capturing()
            self.maps[sid][t] = hidden_states.detach() if capturing else hidden_states.detach().clone()
            if not self.cache_kv:
                return super().__call__(attn, hidden_states, None, attention_mask, temb)
            return super().__call__(attn, hidden_states, None, attention_mask, temb,
                                    kv_sink=lambda k, vt: self.kv[sid].__setitem__(t, (k, vt)))
        if self.cache_kv and t in self.kv[0] and hidden_states.shape[0] % self.kv[0][t][0].shape[0] == 0:
            return super().__call__(attn, hidden_states, None, attention_mask, temb, kv=self.kv[0][t])
        map0 = self._kv_source(attn, self.maps[0][t], hidden_states.shape[0])
        if not self.enable_interp:
            return super().__call__(attn, hidden_states, map0, attention_mask, temb)
        alpha = self.attn_state.alpha
        map1 = self._kv_source(attn, self.maps[1][t], hidden_states.shape[0])
        r1 = super().__call__(attn, hidden_states, map0, attention_mask, temb)
        r2 = super().__call__(attn, hidden_states, map1, attention_mask, temb)
        return (1 - alpha) * r1 + alpha * r2


def get_unet_attn_processors(unet):
    """{'<module path>.processor': processor} for every module exposing get_processor()."""
    found = {}

    def walk(name, module):
        if hasattr(module, "get_processor"):
            found[f"{name}.processor"] = module.get_processor()
        for sub, child in module.named_children():
            walk(f"{name}.{sub}", child)

    for name, module in unet.named_children():
        walk(name, module)
    return found


def set_unet_attn_processor(unet, processor):
    count = len(get_unet_attn_processors(unet))
    if isinstance(processor, dict) and len(processor) != count:
        raise ValueError(
            f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
            f" number of attention layers: {count}. Please make sure to pass {count} processor classes.")

    def walk(name, module):
        if hasattr(module, "set_processor"):
            module.set_processor(processor.pop(f"{name}.processor") if isinstance(processor, dict) else processor)
        for sub, child in module.named_children():
            walk(f"{name}.{sub}", child)

    for name, module in unet.named_children():
        walk(name, module)
