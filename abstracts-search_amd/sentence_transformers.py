"""sentence-transformers-shaped Python mirror of the MI355X encoder.

Drop-in for the `SentenceTransformer` surface the reference reaches through
``sidecar-search build -b 32`` (reference Makefile:65; README.md:60:
``SIDECARSEARCH_MODEL``, ``SIDECARSEARCH_TRUST_REMOTE_CODE``) and its query-time
``app.py`` (reference README.md:28: ``MODEL_NAME``, ``PROMPT_NAME=s2p_query``,
``TRUST_REMOTE_CODE``): the constructor, ``encode``, ``tokenize``,
``get_sentence_embedding_dimension``, ``max_seq_length``, ``prompts``.

The numeric path (embedding gather, 28 decoder layers, pooling, Dense,
normalisation) runs in HIP kernels behind ``include/mi_encoder.h``; this file
tokenises on the host (the ``tokenizers`` library, as sentence-transformers
does), sorts by length, packs batches without padding tokens and moves
pointers.  There is no CPU fallback.

A model directory in the usual sentence-transformers layout is read directly:
``config.json``, ``*.safetensors`` (HF Qwen2 names), ``tokenizer.json``,
``modules.json`` + ``<n>_Dense*/`` (``config.json``, ``model.safetensors``),
``config_sentence_transformers.json`` (prompts), ``sentence_bert_config.json``
(max_seq_length).  Without a checkpoint (this build environment has none), pass
``config=`` and ``weights=`` (name -> tensor) explicitly.
"""
from __future__ import annotations

import ctypes
import json
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

import numpy as np

from . import _native


class _Cfg(ctypes.Structure):
    _fields_ = [("vocab_size", c_int32), ("hidden", c_int32), ("n_layers", c_int32), ("n_heads", c_int32),
                ("n_kv_heads", c_int32), ("head_dim", c_int32), ("intermediate", c_int32),
                ("rms_eps", c_float), ("rope_theta", c_float), ("causal", c_int32), ("dense_out", c_int32),
                ("dense_bias", c_int32), ("max_seq_len", c_int32)]


class _Lib:
    _lib = None

    @classmethod
    def get(cls):
        if cls._lib is None:
            lib = _native.load("encoder")
            lib.mi_enc_last_error.restype = c_char_p
            v = c_void_p
            sigs = {
                "mi_encoder_create": [POINTER(_Cfg), c_int, POINTER(v)],
                "mi_encoder_destroy": [v],
                "mi_encoder_load_tensor": [v, c_char_p, v, c_int, POINTER(c_int64), c_int],
                "mi_encoder_missing": [v, POINTER(c_int)],
                "mi_encoder_out_dim": [v, POINTER(c_int)],
                "mi_encoder_encode": [v, c_int, v, v, c_int, v, v],
                "mi_encoder_encode_rows": [v, c_int, v, v, c_int, v, v, v],
                "mi_encoder_hidden": [v, c_int, v, v, v, v],
                "mi_encoder_profile_enable": [v, c_int],
                "mi_encoder_profile_read": [v, POINTER(c_double), POINTER(c_double)],
                "mi_enc_gemm_bf16": [c_int, c_int, c_int, c_int, v, v, v, v],
                "mi_enc_debug_counter": [c_char_p, POINTER(c_int64)],
                "mi_encoder_reload_env": [],
            }
            for name, args in sigs.items():
                fn = getattr(lib, name)
                fn.argtypes = args
                fn.restype = c_int
            cls._lib = lib
        return cls._lib


def _check(rc: int):
    if rc != 0:
        raise RuntimeError("mi_encoder: " + _Lib.get().mi_enc_last_error().decode())


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


_DTYPES = {"float32": 0, "bfloat16": 1, "float16": 2}

# stella_en_1.5B_v5 (SURVEY Appendix B.1 [PRIOR]; every value is overridable by config.json)
STELLA_EN_1_5B_V5 = dict(vocab_size=151646, hidden=1536, n_layers=28, n_heads=12, n_kv_heads=2, head_dim=128,
                         intermediate=8960, rms_eps=1e-6, rope_theta=1e6, causal=False, dense_out=1024,
                         dense_bias=True, max_seq_len=512)


_ENV_TRUE = ("1", "true", "yes", "on")
_ENV_FALSE = ("0", "false", "no", "off")


def _causal_from_hf(hf: dict, is_embedding_model: bool = False, override=None) -> bool:
    """Which attention mask a checkpoint's config.json asks for -- never guessed silently.

    * an explicit `causal=` (constructor) or MI_ENCODER_CAUSAL=0|1 (also true/false, yes/no, on/off; anything else
      raises) -- for pipelines that cannot pass keywords: `sidecar-search build`, reference Makefile:65 -- wins;
    * `is_causal` in config.json (the key of the gte-Qwen2 remote code) is taken as written;
    * no key and no remote code (`auto_map` absent): a plain Qwen2 checkpoint, which transformers runs causally --
      also under sentence-transformers (modules.json), whose Transformer module calls the same Qwen2Model;
    * no key, `auto_map` names `modeling_qwen.Qwen2Model` AND the directory is a sentence-transformers embedding
      model (modules.json): the published signature of NovaSearch/stella_en_1.5B_v5 and the gte-Qwen2 family it
      derives from, whose remote code runs bidirectional attention [PRIOR: SURVEY App. B.1; the file could not be
      fetched here].  Taken as bidirectional WITH a warning, so that the reference's own command lines
      (README.md:28, README.md:60) load without an extra knob;
    * any other remote modelling code without the key: the mask is decided inside a modeling_*.py this package does
      not execute.  Running such a model with the wrong mask gives plausible, wrong embeddings and no error, so this
      raises and asks."""
    if override is None:
        env = os.environ.get("MI_ENCODER_CAUSAL", "").strip().lower()
        if env != "":
            if env in _ENV_TRUE:
                override = True
            elif env in _ENV_FALSE:
                override = False
            else:
                raise ValueError(f"MI_ENCODER_CAUSAL={env!r}: expected one of {_ENV_FALSE + _ENV_TRUE}")
    if override is not None:
        return bool(override)
    if "is_causal" in hf:
        return bool(hf["is_causal"])
    auto_map = hf.get("auto_map") or {}
    if auto_map:
        targets = sorted(set(str(v) for v in auto_map.values()))
        if is_embedding_model and str(auto_map.get("AutoModel", "")).split("--")[-1] == "modeling_qwen.Qwen2Model":
            import warnings
            warnings.warn(
                "config.json has no `is_causal` key; its auto_map (modeling_qwen.Qwen2Model) and modules.json match the "
                "stella_en_1.5B_v5 / gte-Qwen2 embedding models, whose remote code attends bidirectionally: running "
                "BIDIRECTIONAL attention.  Pass causal=True/False or set MI_ENCODER_CAUSAL=1/0 to decide explicitly.",
                stacklevel=3)
            return False
        raise ValueError(
            "config.json has no `is_causal` key and its `auto_map` points at remote modelling code "
            f"({targets}): whether attention is causal or "
            "bidirectional is decided inside that code, which is not executed here.  Pass "
            "SentenceTransformer(..., causal=False) for a bidirectional encoder (stella / gte-Qwen2 style) or "
            "causal=True, or set MI_ENCODER_CAUSAL=0|1" +
            (" (this directory is a sentence-transformers embedding model: modules.json present)"
             if is_embedding_model else ""))
    return True


def _cfg_from_hf(hf: dict, is_embedding_model: bool = False, causal=None) -> dict:
    hidden = int(hf["hidden_size"])
    nh = int(hf["num_attention_heads"])
    return dict(vocab_size=int(hf["vocab_size"]), hidden=hidden, n_layers=int(hf["num_hidden_layers"]),
                n_heads=nh, n_kv_heads=int(hf.get("num_key_value_heads", nh)),
                head_dim=int(hf.get("head_dim") or hidden // nh), intermediate=int(hf["intermediate_size"]),
                rms_eps=float(hf.get("rms_norm_eps", 1e-6)), rope_theta=float(hf.get("rope_theta", 1e6)),
                causal=_causal_from_hf(hf, is_embedding_model, causal), dense_out=0, dense_bias=True,
                max_seq_len=int(hf.get("max_position_embeddings", 512)))


class HostTokenizer:
    """Host-side tokenisation with sentence-transformers / HF semantics (the
    `tokenizers` library does the work, nothing is re-implemented): `truncation=True,
    max_length=max_seq_length` keeps the special tokens the post-processor adds (the tail of
    an over-long input is cut *before* post-processing, as HF fast tokenizers do), and a
    tokenizer_config.json with `add_eos_token: true` whose tokenizer.json does not already
    append the end-of-sequence token gets it appended after truncating to max_length - 1 --
    what a trust_remote_code tokenizer class that appends EOS at run time does."""

    def __init__(self, tokenizer, add_eos: bool = False, eos_token: str | None = None):
        self.tk = tokenizer
        self.eos_id = tokenizer.token_to_id(eos_token) if (add_eos and eos_token) else None
        self.appends_eos = False
        if self.eos_id is not None:
            probe = tokenizer.encode("a").ids
            self.appends_eos = bool(probe) and probe[-1] == self.eos_id
        self._trunc = None
        try:
            tokenizer.no_padding()
        except Exception:
            pass

    @classmethod
    def from_dir(cls, path: str, tokenizer=None):
        from tokenizers import Tokenizer
        tk = tokenizer if tokenizer is not None else Tokenizer.from_file(os.path.join(path, "tokenizer.json"))
        add_eos, eos = False, None
        tc = os.path.join(path, "tokenizer_config.json")
        if os.path.exists(tc):
            c = json.load(open(tc))
            add_eos = bool(c.get("add_eos_token", False))
            eos = c.get("eos_token")
            if isinstance(eos, dict):
                eos = eos.get("content")
        return cls(tk, add_eos, eos)

    def __call__(self, texts, max_length: int):
        manual_eos = self.eos_id is not None and not self.appends_eos
        want = max(1, max_length - 1) if manual_eos else max_length
        if self._trunc != want:
            self.tk.enable_truncation(max_length=want)
            self._trunc = want
        out = []
        for e in self.tk.encode_batch(list(texts)):
            ids = list(e.ids)
            if manual_eos:
                ids.append(self.eos_id)
            out.append(ids if ids else [0])
        return out


class SentenceTransformer:
    """sentence_transformers.SentenceTransformer on one MI355X."""

    def __init__(self, model_name_or_path: str | None = None, device=None, prompts: dict | None = None,
                 default_prompt_name: str | None = None, trust_remote_code: bool = False,
                 config: dict | None = None, weights: dict | None = None, tokenizer=None,
                 max_seq_length: int | None = None, causal: bool | None = None, **_ignored):
        self.trust_remote_code = trust_remote_code
        mk = _ignored.get("model_kwargs") or {}
        if causal is None and "is_causal" in mk:                 # sentence-transformers' way of reaching the HF config
            causal = bool(mk["is_causal"])
        self._causal_override = causal
        self.prompts = dict(prompts or {})
        self.default_prompt_name = default_prompt_name
        self.tokenizer = tokenizer
        self._host_tok = HostTokenizer(tokenizer) if tokenizer is not None else None
        self._device_index = 0
        if device is not None:
            s = str(device)
            self._device_index = int(s.split(":")[1]) if ":" in s else 0
        cfg = dict(config) if config is not None else None
        if model_name_or_path is not None and not os.path.isdir(model_name_or_path) and cfg is None:
            model_name_or_path = self._resolve_hub_id(model_name_or_path, _ignored)
        if model_name_or_path is not None and os.path.isdir(model_name_or_path):
            cfg, weights = self._read_model_dir(model_name_or_path, cfg, weights)
        if cfg is None:
            raise ValueError("SentenceTransformer: need a model directory or config=")
        if hasattr(cfg, "to_dict"):
            cfg = cfg.to_dict()
        if max_seq_length is not None:
            cfg["max_seq_len"] = int(max_seq_length)
        if causal is not None:
            cfg["causal"] = bool(causal)
        self.config = cfg
        c = _Cfg(**{k: (int(v) if isinstance(v, bool) else v) for k, v in cfg.items() if k in dict(_Cfg._fields_)})
        self._h = c_void_p()
        _check(_Lib.get().mi_encoder_create(ctypes.byref(c), self._device_index, ctypes.byref(self._h)))
        if weights is not None:
            self.load_weights(weights)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _Lib.get().mi_encoder_destroy(h)
            except Exception:
                pass

    # -- loading -------------------------------------------------------
    @staticmethod
    def _resolve_hub_id(name: str, kw: dict) -> str:
        """A Hugging Face hub id the way the reference names its model (README.md:28
        MODEL_NAME="NovaSearch/stella_en_1.5B_v5", README.md:60 SIDECARSEARCH_MODEL=...) -> the snapshot directory:
        the local hub cache first (HF_HOME / HF_HUB_CACHE / cache_folder=, exactly sentence-transformers' lookup),
        then the network unless HF_HUB_OFFLINE is set.  No silent fallback: a model that cannot be found is an error
        that says where it was looked for."""
        try:
            from huggingface_hub import snapshot_download
        except ImportError as e:                                           # pragma: no cover
            raise FileNotFoundError(f"{name!r} is not a local model directory and huggingface_hub is not installed") from e
        common = dict(repo_id=name, cache_dir=kw.get("cache_folder"), revision=kw.get("revision"),
                      token=kw.get("token", kw.get("use_auth_token")))
        try:
            return snapshot_download(local_files_only=True, **common)
        except Exception as local_err:
            if os.environ.get("HF_HUB_OFFLINE", "").strip() not in ("", "0") or kw.get("local_files_only"):
                raise FileNotFoundError(
                    f"{name!r} is neither a local model directory nor in the Hugging Face cache "
                    f"(HF_HOME={os.environ.get('HF_HOME', '~/.cache/huggingface')}, cache_folder={kw.get('cache_folder')}) "
                    f"and the hub is offline: {local_err}") from local_err
            try:
                return snapshot_download(**common)
            except Exception as net_err:
                raise FileNotFoundError(
                    f"{name!r} is neither a local model directory nor in the Hugging Face cache, and the download "
                    f"failed: {net_err}") from net_err

    def _read_model_dir(self, path, cfg, weights):
        from safetensors import safe_open
        with open(os.path.join(path, "config.json")) as f:
            hf = json.load(f)
        cfg = cfg or _cfg_from_hf(hf, os.path.exists(os.path.join(path, "modules.json")), self._causal_override)
        weights = dict(weights or {})
        files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        for fn in files:
            with safe_open(os.path.join(path, fn), framework="pt") as sf:
                for k in sf.keys():
                    weights[k[6:] if k.startswith("model.") else k] = sf.get_tensor(k)
        mods = os.path.join(path, "modules.json")
        if os.path.exists(mods):
            for m in json.load(open(mods)):
                if m.get("type", "").endswith("Dense"):
                    dd = os.path.join(path, m["path"])
                    dc = json.load(open(os.path.join(dd, "config.json")))
                    cfg["dense_out"] = int(dc["out_features"])
                    cfg["dense_bias"] = bool(dc.get("bias", True))
                    with safe_open(os.path.join(dd, "model.safetensors"), framework="pt") as sf:
                        weights["dense.weight"] = sf.get_tensor("linear.weight")
                        if cfg["dense_bias"]:
                            weights["dense.bias"] = sf.get_tensor("linear.bias")
        st_cfg = os.path.join(path, "config_sentence_transformers.json")
        if os.path.exists(st_cfg):
            sc = json.load(open(st_cfg))
            self.prompts = {**sc.get("prompts", {}), **self.prompts}
            self.default_prompt_name = self.default_prompt_name or sc.get("default_prompt_name")
        sb = os.path.join(path, "sentence_bert_config.json")
        if os.path.exists(sb):
            cfg["max_seq_len"] = int(json.load(open(sb)).get("max_seq_length", cfg["max_seq_len"]))
        tk = os.path.join(path, "tokenizer.json")
        if self.tokenizer is not None or os.path.exists(tk):
            self._host_tok = HostTokenizer.from_dir(path, self.tokenizer)   # + tokenizer_config.json (add_eos_token)
            self.tokenizer = self._host_tok.tk
        return cfg, weights

    def load_weights(self, weights: dict):
        """name -> torch tensor / numpy array (f32, bf16 or f16; host or CUDA)."""
        import torch
        lib = _Lib.get()
        for name, t in weights.items():
            if name.startswith("model."):
                name = name[6:]
            if name == "lm_head.weight" or "rotary_emb" in name:
                continue
            if not _is_torch(t):
                t = torch.from_numpy(np.ascontiguousarray(t))
            t = t.contiguous()
            dt = str(t.dtype).replace("torch.", "")
            if dt not in _DTYPES:
                t, dt = t.float(), "float32"
            shape = (c_int64 * t.dim())(*t.shape)
            _check(lib.mi_encoder_load_tensor(self._h, name.encode(), c_void_p(t.data_ptr()), _DTYPES[dt],
                                              shape, t.dim()))

    # -- sentence-transformers surface ----------------------------------
    @property
    def max_seq_length(self) -> int:
        return int(self.config["max_seq_len"])

    def get_sentence_embedding_dimension(self) -> int:
        n = c_int(0)
        _check(_Lib.get().mi_encoder_out_dim(self._h, ctypes.byref(n)))
        return n.value

    def get_max_seq_length(self) -> int:
        return self.max_seq_length

    def tokenize(self, texts):
        """list[str] -> list of token-id lists, truncated to max_seq_length the way
        sentence-transformers does (special tokens survive; see HostTokenizer)."""
        if self._host_tok is None:
            raise RuntimeError("no tokenizer: pass tokenizer= or a model directory with tokenizer.json")
        return self._host_tok(texts, self.max_seq_length)

    def encode(self, sentences, prompt_name: str | None = None, prompt: str | None = None,
               batch_size: int | None = None, show_progress_bar=None, output_value: str | None = "sentence_embedding",
               precision: str = "float32", convert_to_numpy: bool = True, convert_to_tensor: bool = False,
               device=None, normalize_embeddings: bool = False, **_ignored):
        """batch_size: sentence-transformers' default is 32; here None (the default) leaves the size of a forward pass to
        `token_budget`, and a value the caller PASSES bounds a pass from above (at most that many sequences, and at most
        token_budget tokens) -- `sidecar-search build -b 32` (reference Makefile:65) therefore means 32, as it says."""
        if output_value not in (None, "sentence_embedding") or precision != "float32":
            raise NotImplementedError("only sentence embeddings in float32 are implemented "
                                      f"(output_value={output_value!r}, precision={precision!r})")
        single = isinstance(sentences, str)
        if single:
            sentences = [sentences]
        if prompt is None:
            name = prompt_name or self.default_prompt_name
            if name is not None:
                if name not in self.prompts:
                    raise ValueError(f"Prompt name '{name}' not found in the configured prompts dictionary "
                                     f"with keys {list(self.prompts)!r}.")
                prompt = self.prompts[name]
        if prompt:
            sentences = [prompt + s for s in sentences]
        toks = self.tokenize(sentences)
        emb = self.encode_tokens(toks, batch_size=batch_size, normalize_embeddings=normalize_embeddings,
                                 as_tensor=convert_to_tensor)
        if single:
            emb = emb[0]
        return emb

    # Tokens one forward pass takes: 32 768 = 128 row tiles of 256, with which every GEMM of the stack fills whole
    # rounds of the 256 CUs (N = 1536: 768 tiles = 3.0 rounds, 2048: 4.0, 17 920: 35.0).  sentence-transformers'
    # batch_size (32 by default) is a memory knob of its hardware; the token budget is the memory bound here
    # (activations of 32 768 tokens are ~1.2 GB).  A batch_size the caller passes explicitly still bounds a pass from
    # above (`-b 32` in the reference's Makefile:65 means 32); encode()'s default (None) leaves it to the budget.
    # Embeddings do not depend on which sequences share a pass beyond bf16 / f32 rounding order (tests: batching
    # invariance).  `token_budget = None`: passes of exactly `batch_size` length-sorted sequences.
    token_budget = 32768

    def _passes(self, order, token_lists, batch_size):
        """the sorted sequence indices cut into forward passes: at most `batch_size` sequences (None: no bound) and
        at most `token_budget` tokens (None: no bound) each, never empty"""
        if not self.token_budget:
            bs = batch_size or 32
            return [order[b0:b0 + bs] for b0 in range(0, len(order), bs)]
        passes, cur, tok = [], [], 0
        for i in order:
            t = len(token_lists[i])                               # sequences are packed back to back
            if cur and (tok + t > self.token_budget or (batch_size and len(cur) >= batch_size)):
                passes.append(cur)
                cur, tok = [], 0
            cur.append(i)
            tok += t
        if cur:
            passes.append(cur)
        return passes

    def encode_tokens(self, token_lists, batch_size: int | None = None, normalize_embeddings: bool = False,
                      as_tensor: bool = False):
        """list of token-id lists -> float32 [n, dim].  Like sentence-transformers,
        inputs are sorted by length (longest first) before batching and the
        result is put back in input order; a batch is packed, not padded, and holds at most
        `token_budget` tokens (see above) and at most `batch_size` sequences (None: no bound)."""
        import itertools
        import torch
        n = len(token_lists)
        dim = self.get_sentence_embedding_dimension()
        dev = torch.device("cuda", self._device_index)
        out = torch.empty((n, dim), dtype=torch.float32, device=dev)
        order = sorted(range(n), key=lambda i: -len(token_lists[i]))
        stream = c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        lib = _Lib.get()
        # every pass only ENQUEUES work (the C ABI stages the token ids through pinned memory and writes embedding i
        # straight to row sel[i] of `out`): the packing of the next pass overlaps the GPU running this one
        for sel in self._passes(order, token_lists, batch_size):
            cu = np.zeros(len(sel) + 1, np.int32)
            np.cumsum([len(token_lists[i]) for i in sel], out=cu[1:])
            if isinstance(token_lists[sel[0]], np.ndarray):           # id arrays (what `tokenizers` can hand over)
                ids = np.concatenate([token_lists[i] for i in sel]).astype(np.int32, copy=False)
            else:
                ids = np.fromiter(itertools.chain.from_iterable(token_lists[i] for i in sel), np.int32, count=int(cu[-1]))
            rows = np.asarray(sel, np.int32)
            _check(lib.mi_encoder_encode_rows(self._h, len(sel), c_void_p(ids.ctypes.data), c_void_p(cu.ctypes.data),
                                              int(normalize_embeddings), c_void_p(out.data_ptr()), c_void_p(rows.ctypes.data),
                                              stream))
        if as_tensor:
            return out
        return out.cpu().numpy()

    # -- parity / measurement hooks ---------------------------------------
    def last_hidden_state(self, token_lists):
        """packed float32 [T, hidden] after the final norm (tests)."""
        lens = [len(t) for t in token_lists]
        cu = np.zeros(len(lens) + 1, np.int32)
        np.cumsum(lens, out=cu[1:])
        ids = np.fromiter((t for tl in token_lists for t in tl), np.int32, count=int(cu[-1]))
        out = np.empty((int(cu[-1]), int(self.config["hidden"])), np.float32)
        _check(_Lib.get().mi_encoder_hidden(self._h, len(lens), c_void_p(ids.ctypes.data), c_void_p(cu.ctypes.data),
                                            c_void_p(out.ctypes.data), c_void_p(0)))
        return out

    def profile(self, on: bool = True):
        _check(_Lib.get().mi_encoder_profile_enable(self._h, int(on)))

    def profile_read(self):
        ms, fl = c_double(0), c_double(0)
        _check(_Lib.get().mi_encoder_profile_read(self._h, ctypes.byref(ms), ctypes.byref(fl)))
        return {"gemm_ms": ms.value, "gemm_flops": fl.value}


def reload_env() -> None:
    """re-read the library's MI_* knobs from the environment (tests / tools; the library reads them once otherwise)"""
    _check(_Lib.get().mi_encoder_reload_env())


def debug_counter(name: str) -> int:
    """process-wide counters of dispatcher decisions (tests): 'tail_split_launches'"""
    v = c_int64(0)
    _check(_Lib.get().mi_enc_debug_counter(name.encode(), ctypes.byref(v)))
    return v.value


def gemm_bf16(A, W):
    """C = A @ W.T in bf16 with f32 accumulation on the MFMA kernel (tests)."""
    import torch
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty((M, N), dtype=torch.bfloat16, device=A.device)
    _check(_Lib.get().mi_enc_gemm_bf16(A.device.index or 0, M, N, K, c_void_p(A.data_ptr()),
                                       c_void_p(W.data_ptr()), c_void_p(C.data_ptr()),
                                       c_void_p(torch.cuda.current_stream().cuda_stream)))
    return C
