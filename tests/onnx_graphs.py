"""Asset directories whose .onnx files hold real node graphs of the engine's layout (test infrastructure): every stage is
emitted in execution order the way an exporter would — weightless nodes (Transpose, Gelu, Softmax, Reshape, residual Adds)
between the weighted ones, projections in four encodings (Conv k=1, MatMul + Add, Gemm transB=1, Gemm transB=0), shape constants
as initializers or Constant nodes, weights as raw / packed / fp64 storage — and NO manifest.  stn_load_dir has to recognise the
layout from the nodes (supertonic_amd/csrc/host/graph_bind.cpp).  `tensor(name)` supplies the flat canonical fp32 weights."""
import json

import numpy as np

import onnx_writer as ow

FILES = {"dp": "duration_predictor.onnx", "te": "text_encoder.onnx", "ve": "vector_estimator.onnx", "vo": "vocoder.onnx"}
IO = {"dp": (["text_ids", "style_dp", "text_mask"], ["duration"]), "te": (["text_ids", "style_ttl", "text_mask"], ["text_emb"]),
      "ve": (["noisy_latent", "text_emb", "style_ttl", "text_mask", "latent_mask", "total_step", "current_step"], ["denoised_latent"]),
      "vo": (["latent"], ["wav_tts"])}


class Graph:
    """variants (exporter spellings of the same layout, all must bind to the same tensors):
         ln   = "node" | "decomposed" | "decomposed_nobeta"   LayerNormalization as one node, or ReduceMean / Sub / Pow / ReduceMean / Add eps /
                                                              Sqrt / Div / Mul gamma / Add beta (beta dropped in the last form)
         qkv  = "separate" | "fused"                          q, k, v projections separately, or one 3C (self) / 2C (context k|v) projection + Split
         cut  = "split" | "slices" | "swapped" | "axis1" | "none"   (qkv = "fused") how the fused projection's result is divided: one Split on the last axis, one
                                                              Slice per part, a Split whose outputs reach the attention as k, q, v / v, k, a Split on the
                                                              wrong axis, or no cut node at all (the last three are NOT the layout: the loader must say so)
         pw   = "mixed" | "matmul_transpose"                  projections in four encodings by position, or always Transpose -> MatMul -> Add -> Transpose
         gelu = "op" | "op_tanh" | "erf" | "tanh"             a Gelu node (approximate = none / tanh), or the Erf / Tanh formulas spelled out
         head = "linear" | "convtranspose" | "convtranspose_overlap"
                                                              the vocoder's wave head as a projection, or as a one-channel ConvTranspose with stride ==
                                                              kernel == base_chunk_size (the same arithmetic: one bias value for the whole frame), or with
                                                              kernel == 2 x stride (an overlap-add head: NOT the layout, the loader must say so)"""

    def __init__(self, stage, tensor, breaks, variants=None):
        self.stage, self.tensor, self.breaks = stage, tensor, breaks
        self.var = dict(ln="node", qkv="separate", pw="mixed", gelu="op", head="linear", cut="split")
        self.var.update(variants or {})
        self.inits, self.nodes, self.n = [], [], 0
        self.cur = IO[stage][0][0]

    def scalar(self, v):
        return self.init("scalar", np.array(v, np.float32).reshape(()))

    def gelu(self):
        g, x = self.var["gelu"], self.cur
        if g == "op":
            self.plain("Gelu")
        elif g == "op_tanh":
            self.cur = self.op("Gelu", [x], [ow.attr_str("approximate", "tanh")])
        elif g == "erf":
            e = self.op("Erf", [self.op("Div", [x, self.scalar(np.sqrt(2.0))])])
            self.cur = self.op("Mul", [self.op("Mul", [x, self.op("Add", [e, self.scalar(1.0)])]), self.scalar(0.5)])
        else:
            x3 = self.op("Pow", [x, self.scalar(3.0)])
            t = self.op("Tanh", [self.op("Mul", [self.op("Add", [x, self.op("Mul", [x3, self.scalar(0.044715)])]), self.scalar(np.sqrt(2.0 / np.pi))])])
            self.cur = self.op("Mul", [self.op("Mul", [x, self.op("Add", [t, self.scalar(1.0)])]), self.scalar(0.5)])

    def val(self):
        self.n += 1
        return f"/{self.stage}/v{self.n}"

    def init(self, name, arr):
        iname = f"onnx::{self.stage}_{len(self.inits)}"  # exporter-style names: nothing to match by
        style = ("raw", "packed", "raw", "f64")[len(self.inits) % 4]
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.float32 and style == "f64":
            self.inits.append(ow.tensor(iname, arr.astype(np.float64)))
        elif arr.dtype == np.float32 and style == "packed":
            self.inits.append(ow.tensor(iname, arr, style="packed"))
        else:
            self.inits.append(ow.tensor(iname, arr))
        return iname

    def op(self, op_type, inputs, attrs=(), name=None):
        out = self.val()
        self.nodes.append(ow.node(op_type, inputs, [out], name or f"/{self.stage}/{op_type}_{len(self.nodes)}", attrs))
        return out

    def plain(self, op_type, *more):
        self.cur = self.op(op_type, [self.cur, *more])

    def w(self, name, shape):
        return self.tensor(name).reshape(shape)

    def embed(self, name, V, C):
        self.cur = self.op("Gather", [self.init(name, self.w(name, (V, C))), self.cur], [ow.attr_int("axis", 0)])

    def linear(self, name, out, inp, src=None):
        x = src if src is not None else self.cur
        W, b = self.w(name + ".w", (out, inp)), self.w(name + ".b", (out,))
        return self.linear_wb(name, W, b, x, src is None)

    def linear_wb(self, name, W, b, x, advance=True):
        out, inp = W.shape
        enc = len(self.nodes) % 4
        if self.var["pw"] == "matmul_transpose":  # what a Conv1d(k=1) becomes when an exporter lowers it: Transpose -> MatMul -> Add -> Transpose
            y = self.op("Transpose", [x], [ow.attr_ints("perm", [0, 2, 1])])
            y = self.op("MatMul", [y, self.init(name, W.T)])
            y = self.op("Add", [y, self.init(name, b)])
            y = self.op("Transpose", [y], [ow.attr_ints("perm", [0, 2, 1])])
        elif enc == 0:
            y = self.op("Conv", [x, self.init(name, W.reshape(out, inp, 1)), self.init(name, b)], [ow.attr_ints("kernel_shape", [1])])
        elif enc == 1:
            y = self.op("MatMul", [x, self.init(name, W.T)])
            y = self.op("Add", [self.init(name, b), y] if len(self.nodes) % 3 == 0 else [y, self.init(name, b)])
        elif enc == 2:
            y = self.op("Gemm", [x, self.init(name, W), self.init(name, b)], [ow.attr_int("transB", 1)])
        else:
            y = self.op("Gemm", [x, self.init(name, W.T), self.init(name, b)])
        if advance:
            self.cur = y
        return y

    def ln(self, name, C):
        if self.var["ln"] == "node":
            self.cur = self.op("LayerNormalization", [self.cur, self.init(name, self.w(name + ".g", (C,))), self.init(name, self.w(name + ".b", (C,)))],
                               [ow.attr_int("axis", -1)])
            return
        x = self.cur
        mu = self.op("ReduceMean", [x], [ow.attr_ints("axes", [-1])])
        d = self.op("Sub", [x, mu])
        var = self.op("ReduceMean", [self.op("Pow", [d, self.scalar(2.0)])], [ow.attr_ints("axes", [-1])])
        xh = self.op("Div", [d, self.op("Sqrt", [self.op("Add", [var, self.scalar(1e-6)])])])
        y = self.op("Mul", [xh, self.init(name, self.w(name + ".g", (C,)))] if len(self.nodes) % 2 else [self.init(name, self.w(name + ".g", (C,))), xh])
        if self.var["ln"] != "decomposed_nobeta":
            y = self.op("Add", [y, self.init(name, self.w(name + ".b", (C,)))])
        self.cur = y

    def convnext(self, name, C, H, k, dil):
        res = self.cur
        brk = self.breaks.get(name)
        Cw = C + 8 if brk == "width" else C
        W = self.w(name + ".dw.w", (C, 1, k))
        if brk == "width":
            W = np.concatenate([W, W[:8]], 0)
        self.cur = self.op("Conv", [self.cur, self.init(name, W), self.init(name, np.resize(self.w(name + ".dw.b", (C,)), Cw))],
                           [ow.attr_ints("dilations", [dil], packed=dil % 2 == 0), ow.attr_int("group", Cw), ow.attr_ints("kernel_shape", [k]),
                            ow.attr_ints("pads", [dil * (k // 2)] * 2)])
        self.plain("Transpose")
        if brk == "width":
            self.cur = self.op("LayerNormalization", [self.cur, self.init(name, np.ones(Cw, np.float32)), self.init(name, np.zeros(Cw, np.float32))])
        else:
            self.ln(name + ".ln", C)
        if brk == "batchnorm":
            self.cur = self.op("BatchNormalization", [self.cur] + [self.init(name, np.ones(C, np.float32)) for _ in range(4)])
        self.linear(name + ".pw1", H, C)
        self.gelu()
        self.linear(name + ".pw2", C, H)
        if brk != "no_gamma":
            g = self.w(name + ".gamma", (C,))
            self.cur = self.op("Mul", [self.cur, self.init(name, g.reshape(1, 1, C) if len(self.nodes) % 2 else g)])
        self.plain("Transpose")
        self.cur = self.op("Add", [res, self.cur])

    def shape_const(self, vals):
        arr = np.array(vals, np.int64)
        if len(self.nodes) % 2:
            return self.init("shape", arr)
        out = self.val()  # a Constant node: the tensor rides in the node's "value" attribute
        self.nodes.append(ow.node("Constant", [], [out], f"/{self.stage}/Constant_{len(self.nodes)}", [ow.attr_tensor("value", ow.tensor("", arr))]))
        return out

    def attn(self, name, C, Cctx, heads, ctx, with_heads=True):
        res = self.cur
        self.ln(name + ".ln", C)
        kv_src = self.cur if ctx is None else ctx
        split = lambda x: self.op("Reshape", [x, self.shape_const([0, 0, heads, C // heads])]) if with_heads else self.op("Identity", [x])
        cat = lambda parts, suf, shp: np.concatenate([self.w(f"{name}.{p_}.{suf}", shp) for p_ in parts], 0)
        def cut(f, names):
            """the fused projection's result f -> one value per name, by the `cut` variant"""
            mode = self.var["cut"]
            outs = [f + "_" + n_ for n_ in names]
            if mode == "slices":  # one Slice per part, emitted in REVERSE order (the binder must go by the offsets, not by node order)
                for j in reversed(range(len(names))):
                    ins = [f] + [self.init("slice", np.array([v_], np.int64)) for v_ in (j * C, (j + 1) * C, -1)]
                    self.nodes.append(ow.node("Slice", ins, [outs[j]], f"/{self.stage}/Slice_{len(self.nodes)}"))
            elif mode == "none":
                return [f] * len(names)
            else:
                o = list(reversed(outs)) if mode == "swapped" and len(outs) == 2 else ([outs[1], outs[0]] + outs[2:] if mode == "swapped" else outs)
                self.nodes.append(ow.node("Split", [f], o, f"/{self.stage}/Split_{len(self.nodes)}", [ow.attr_int("axis", 1 if mode == "axis1" else -1)]))
            return outs

        if self.var["qkv"] == "fused" and ctx is None:  # self-attention: ONE projection of 3C rows, split afterwards
            f = self.linear_wb(name + ".qkv", cat("qkv", "w", (C, C)), cat("qkv", "b", (C,)), self.cur, advance=False)
            q, k, v = (split(x) for x in cut(f, "qkv"))
        elif self.var["qkv"] == "fused":  # cross-attention: q alone, k | v as ONE projection of the context
            q = split(self.linear(name + ".q", C, C, src=self.cur))
            f = self.linear_wb(name + ".kv", cat("kv", "w", (C, Cctx)), cat("kv", "b", (C,)), kv_src, advance=False)
            k, v = (split(x) for x in cut(f, "kv"))
        else:
            q = split(self.linear(name + ".q", C, C, src=self.cur))
            k = split(self.linear(name + ".k", C, Cctx, src=kv_src))
            v = split(self.linear(name + ".v", C, Cctx, src=kv_src))
        s = self.op("MatMul", [self.op("Transpose", [q]), self.op("Transpose", [k])])
        s = self.op("Mul", [s, self.init("scale", np.array([1.0 / np.sqrt(C // heads)], np.float32))])  # a scalar: not a weight
        o = self.op("MatMul", [self.op("Softmax", [s]), self.op("Transpose", [v])])
        self.cur = self.op("Reshape", [self.op("Transpose", [o]), self.shape_const([0, 0, C])])
        self.linear(name + ".o", C, C)
        self.cur = self.op("Add", [res, self.cur])

    def model(self):
        ins, outs = IO[self.stage]
        self.nodes.append(ow.node("Identity", [self.cur], [outs[0]], f"/{self.stage}/out"))
        return ow.model(self.inits, self.nodes, ins, outs, producer="pytorch")


def build_graph_dir(tmp, a, tensor, breaks=None, with_heads=True, tts_overrides=None, io_overrides=None, variants=None):
    """Write tts.json, unicode_indexer.json and the four graphs of descriptor `a` into `tmp` (no manifest).  variants: see Graph."""
    from supertonic_amd import host
    breaks = breaks or {}
    D = a.latent_dim * a.chunk_compress_factor
    g = Graph("dp", tensor, breaks, variants)
    g.embed("dp.emb", a.vocab_size, a.dp_dim)
    for i in range(a.dp_conv_blocks):
        g.convnext(f"dp.conv{i}", a.dp_dim, a.dp_hidden, a.dp_kernel, 1)
    g.attn("dp.st", a.dp_dim, a.d_style_dp, a.dp_heads, "style_dp", with_heads)
    g.ln("dp.out_ln", a.dp_dim)
    g.linear("dp.fc1", a.dp_dim, a.dp_dim)
    g.plain("Relu")
    g.linear("dp.fc2", 1, a.dp_dim)
    graphs = {"dp": g}

    g = Graph("te", tensor, breaks, variants)
    g.embed("te.emb", a.vocab_size, a.te_dim)
    for i in range(a.te_conv_blocks):
        g.convnext(f"te.conv{i}", a.te_dim, a.te_hidden, a.te_kernel, 1)
    for i in range(a.te_attn_blocks):
        g.attn(f"te.sa{i}", a.te_dim, a.te_dim, a.te_heads, None, with_heads)
        res = g.cur
        g.ln(f"te.sa{i}.ffn_ln", a.te_dim)
        g.linear(f"te.sa{i}.ffn1", a.te_ffn, a.te_dim)
        g.gelu()
        g.linear(f"te.sa{i}.ffn2", a.te_dim, a.te_ffn)
        g.cur = g.op("Add", [res, g.cur])
    for i in range(a.te_style_blocks):
        g.attn(f"te.st{i}", a.te_dim, a.d_style_ttl, a.te_heads, "style_ttl", with_heads)
    g.ln("te.out_ln", a.te_dim)
    g.linear("te.proj", a.te_out_dim, a.te_dim)
    graphs["te"] = g

    g = Graph("ve", tensor, breaks, variants)
    g.linear("ve.in", a.ve_dim, D)
    x = g.cur
    t = g.op("Concat", [g.op("Sin", ["current_step"]), g.op("Cos", ["current_step"])])
    t = g.linear("ve.t1", a.ve_dim, a.ve_time_dim, src=t)
    t = g.linear("ve.t2", a.ve_dim, a.ve_dim, src=g.op("Mish", [t]))
    g.cur = x
    for b in range(a.ve_main_blocks):
        for j in range(a.ve_dilated):
            g.convnext(f"ve.m{b}.dil{j}", a.ve_dim, a.ve_hidden, a.ve_kernel, 1 << j)
        tv = g.linear(f"ve.m{b}.time", a.ve_dim, a.ve_dim, src=t)
        g.cur = g.op("Add", [g.cur, tv])
        g.convnext(f"ve.m{b}.cn_a", a.ve_dim, a.ve_hidden, a.ve_kernel, 1)
        g.attn(f"ve.m{b}.text", a.ve_dim, a.te_out_dim, a.ve_heads, "text_emb", with_heads)
        g.convnext(f"ve.m{b}.cn_b", a.ve_dim, a.ve_hidden, a.ve_kernel, 1)
        g.attn(f"ve.m{b}.style", a.ve_dim, a.d_style_ttl, a.ve_heads, "style_ttl", with_heads)
    for j in range(a.ve_tail_blocks):
        g.convnext(f"ve.tail{j}", a.ve_dim, a.ve_hidden, a.ve_kernel, 1)
    g.ln("ve.out_ln", a.ve_dim)
    g.linear("ve.out", D, a.ve_dim)
    graphs["ve"] = g

    g = Graph("vo", tensor, breaks, variants)
    k = a.vo_in_kernel
    g.cur = g.op("Conv", [g.cur, g.init("vo.in", g.w("vo.in.w", (a.vo_dim, a.latent_dim, k))), g.init("vo.in", g.w("vo.in.b", (a.vo_dim,)))],
                 [ow.attr_ints("kernel_shape", [k]), ow.attr_ints("pads", [k // 2] * 2)])
    for i in range(a.vo_blocks):
        g.convnext(f"vo.blk{i}", a.vo_dim, a.vo_hidden, a.vo_kernel, a.vo_dilations[i])
    g.ln("vo.out_ln", a.vo_dim)
    if g.var["head"] == "linear":
        g.linear("vo.head", a.base_chunk_size, a.vo_dim)
    else:
        cs = a.base_chunk_size
        W, b = g.w("vo.head.w", (cs, a.vo_dim)), g.w("vo.head.b", (cs,))
        kk = cs if g.var["head"] == "convtranspose" else 2 * cs
        Wt = np.ascontiguousarray(W.T).reshape(a.vo_dim, 1, cs)  # ConvTranspose weight: [Cin][Cout / group][k]
        if kk != cs:
            Wt = np.concatenate([Wt, Wt], axis=2)
        g.cur = g.op("ConvTranspose", [g.cur, g.init("vo.head", Wt), g.init("vo.head", b[:1])],
                     [ow.attr_ints("kernel_shape", [kk]), ow.attr_ints("strides", [cs]), ow.attr_ints("pads", [0, 0] if kk == cs else [cs // 2, cs // 2])])
    graphs["vo"] = g

    for st, fn in FILES.items():
        data = graphs[st].model()
        if io_overrides and st in io_overrides:
            gg = graphs[st]
            data = ow.model(gg.inits, gg.nodes, *io_overrides[st], producer="pytorch")
        (tmp / fn).write_bytes(data)
    cfg = {"ae": {"sample_rate": a.sample_rate, "base_chunk_size": a.base_chunk_size},
           "ttl": {"chunk_compress_factor": a.chunk_compress_factor, "latent_dim": a.latent_dim,
                   "style_encoder": {"style_token_layer": {"n_style": a.n_style_ttl, "style_value_dim": a.d_style_ttl}},
                   "text_encoder": {"proj_out": {"idim": a.te_dim, "odim": a.te_out_dim}}},
           "dp": {"style_encoder": {"style_token_layer": {"n_style": a.n_style_dp, "style_value_dim": a.d_style_dp}}}}
    for path, v in (tts_overrides or {}).items():
        cur = cfg
        for key in path[:-1]:
            cur = cur[key]
        cur[path[-1]] = v
    (tmp / "tts.json").write_text(json.dumps(cfg))
    (tmp / "unicode_indexer.json").write_text(json.dumps(host.synthetic_indexer().tolist()))
