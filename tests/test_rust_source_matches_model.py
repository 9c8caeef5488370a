"""Ties the Rust sources of integration/rust/base-hip (never compiled in this image) to their Python model
(tests/rust_binding_model.py), which is what the other tests execute: for every modelled function the sequence of gate
calls (batch.gate / batch.mux / bytes_gate / xtime_many / eval_circuit / xor_layer / the trivial-operand rules of
hip.rs), the literal index tables, and the index expressions of the byte arrays are extracted from both texts and
compared.  An edit to one side that changes which gates are submitted, or in which order, fails here.
The gate sequences themselves follow /root/reference/base/src/state.rs:72-215, key_schedule.rs:90-131,
primitive.rs:250-265,349-362."""
import inspect
import os
import re

from tests import rust_binding_model as M

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "integration", "rust", "base-hip", "src")


def rust_fn(file, name):
    """text of `fn name` in file: from its signature to the matching closing brace"""
    text = open(os.path.join(SRC, file)).read()
    m = re.search(r"\bfn %s\b[^{]*\{" % re.escape(name), text)
    assert m, f"{file}: fn {name} not found"
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    body = text[m.end():i]
    return re.sub(r"//[^\n]*", "", body)


def py_fn(obj):
    src = inspect.getsource(obj)
    src = re.sub(r"#[^\n]*", "", src)
    return src.split(":", 1)[1] if src.lstrip().startswith(("def ", "@")) else src


RUST_CALLS = [
    (r"batch\.gate\(Op::(\w+)", lambda m: "gate:" + m.group(1).lower()),
    (r"batch\.gate\(op\b", lambda m: "gate:op"),
    (r"batch\.mux\(", lambda m: "mux"),
    (r"self\.gate\(Op::(\w+),\s*(&?\w+),\s*(&?\w+)\)", lambda m: "rule:%s(%s,%s)" % (m.group(1).lower(), m.group(2).strip("&"), m.group(3).strip("&"))),
    (r"self\.ready\(", lambda m: "ready"),
    (r"self\.file\(", lambda m: "file"),
    (r"bytes_gate\(Op::(\w+)", lambda m: "bytes_gate:" + m.group(1).lower()),
    (r"xtime_many\(", lambda m: "xtime_many"),
    (r"eval_circuit\(", lambda m: "eval_circuit"),
    (r"Self::xor_layer\(", lambda m: "xor_layer"),
    (r"batch\.run\(\)", lambda m: "run"),
]
PY_CALLS = [
    (r"batch\.gate\((AND|OR|XOR|NAND|NOR|XNOR)\b", lambda m: "gate:" + m.group(1).lower()),
    (r"batch\.gate\(op\b", lambda m: "gate:op"),
    (r"batch\.mux\(", lambda m: "mux"),
    (r"self\.gate\((AND|OR),\s*([\w()_]+?),\s*([\w()_]+?)\)(?=\s|$)", lambda m: "rule:%s(%s,%s)" % (m.group(1).lower(), m.group(2), m.group(3))),
    (r"self\._ready\(", lambda m: "ready"),
    (r"self\._file\(", lambda m: "file"),
    (r"bytes_gate\((AND|OR|XOR)\b", lambda m: "bytes_gate:" + m.group(1).lower()),
    (r"xtime_many\(", lambda m: "xtime_many"),
    (r"eval_circuit\(", lambda m: "eval_circuit"),
    (r"self\.xor_layer\(", lambda m: "xor_layer"),
    (r"batch\.run\(\)", lambda m: "run"),
]


def calls(text, table):
    found = []
    for pat, name in table:
        for m in re.finditer(pat, text):
            found.append((m.start(), name(m)))
    return [n for _, n in sorted(found)]


def tables(text):
    """bracketed lists of at least four comma-separated simple terms, whitespace removed"""
    out = []
    for m in re.finditer(r"\[([^\[\]{};|]*)\]", text):
        items = [re.sub(r"\s+", "", x) for x in m.group(1).split(",")]
        if len(items) >= 4 and all(re.fullmatch(r"[\w+\-*/%]+", x) for x in items):
            out.append(items)
    return out


def index_exprs(text, names):
    """(array, index expression) for every indexing of the named arrays, in order; // is Python's spelling of Rust's /"""
    out = []
    for m in re.finditer(r"\b(?:self\.)?(%s)\[([^\[\]:]+?)\]" % "|".join(names), text):
        if ".." in m.group(2):
            continue
        out.append((m.group(1), re.sub(r"\s+", "", m.group(2)).replace("//", "/")))
    return out


PAIRS = [
    # (rust file, rust fn, python object)
    ("primitive.rs", "eval_circuit", M.eval_circuit),
    ("primitive.rs", "bytes_gate", M.bytes_gate),
    ("primitive.rs", "xtime_many", M.xtime_many),
    ("state.rs", "xor_layer", M.State.xor_layer),
    ("state.rs", "mix_columns", M.State.mix_columns),
    ("state.rs", "inv_mix_columns", M.State.inv_mix_columns),
    ("state.rs", "xor_key_enc", M.State.xor_key_enc),
    ("key_schedule.rs", "generate_next_key_in_place", M.Key.generate_next_key_in_place),
    ("hip.rs", "gate", M.StageBatch.gate),
    ("hip.rs", "mux", M.StageBatch.mux),
]


def test_gate_call_sequences_agree():
    for file, name, obj in PAIRS:
        if file == "hip.rs":          # two `fn gate` / `fn mux` exist (ServerKey's single-gate forms come first): take StageBatch's
            text = open(os.path.join(SRC, file)).read()
            impl = text[text.index("impl<'a> StageBatch<'a>"):]
            m = re.search(r"\bpub fn %s\b[^{]*\{" % name, impl)
            depth, i = 1, m.end()
            while depth:
                depth += {"{": 1, "}": -1}.get(impl[i], 0)
                i += 1
            rust = re.sub(r"//[^\n]*", "", impl[m.end():i])
        else:
            rust = rust_fn(file, name)
        r, p = calls(rust, RUST_CALLS), calls(py_fn(obj), PY_CALLS)
        # the model writes `not_(c)` where the Rust binds `nc = self.key.not(c)` first
        p = [x.replace("not_(c)", "nc") for x in p]
        assert r == p, f"{file}::{name}: Rust {r} != model {p}"
        assert r, f"{file}::{name}: nothing extracted"


def test_index_tables_agree():
    for file, name, obj in PAIRS:
        if file == "hip.rs":
            continue
        r, p = tables(rust_fn(file, name)), tables(py_fn(obj))
        assert r == p, f"{file}::{name}: Rust tables {r} != model tables {p}"
    assert tables(rust_fn("state.rs", "mix_columns")) == [
        "8 9 10 11 8 9 10 11 0 1 2 3 0 1 2 3".split(), "4 5 6 7 0 1 2 3 12 13 14 15 8 9 10 11".split(), "4 5 6 7 8 9 10 11 12 13 14 15 0 1 2 3".split()]
    assert tables(rust_fn("key_schedule.rs", "generate_next_key_in_place")) == [["7", "11", "15", "3"], ["0", "4", "8", "12"], ["c+1", "c+5", "c+9", "c+13"]]
    swaps = re.search(r"const SWAPS[^=]*=\s*\[(.*?)\];", open(os.path.join(SRC, "state.rs")).read(), re.S).group(1)
    assert [tuple(int(x) for x in pr) for pr in re.findall(r"\((\d+),\s*(\d+)\)", swaps)] == M.SWAPS
    swaps = re.search(r"const SWAPS[^=]*=\s*\[(.*?)\];", open(os.path.join(SRC, "key_schedule.rs")).read(), re.S).group(1)
    assert [tuple(int(x) for x in pr) for pr in re.findall(r"\((\d+),\s*(\d+)\)", swaps)] == M.SWAPS
    rcon = re.search(r"const RCON[^=]*=\s*\[(.*?)\];", open(os.path.join(SRC, "key_schedule.rs")).read(), re.S).group(1)
    assert [int(x, 16) for x in re.findall(r"0x[0-9A-Fa-f]+", rcon)] == M.RCON


def test_byte_array_indexing_agrees():
    names = ["d", "y", "t0", "t1", "x", "temp", "data", "rhs", "doubled"]
    for file, name, obj in [("state.rs", "mix_columns", M.State.mix_columns), ("state.rs", "inv_mix_columns", M.State.inv_mix_columns),
                            ("state.rs", "xor_layer", M.State.xor_layer), ("key_schedule.rs", "generate_next_key_in_place", M.Key.generate_next_key_in_place)]:
        r, p = index_exprs(rust_fn(file, name), names), index_exprs(py_fn(obj), names)
        assert r == p, f"{file}::{name}: Rust {r} != model {p}"
        assert r


def test_modes_call_order_is_the_reference_one():
    """ecb_encrypt / ecb_decrypt of the model are the unchanged callers of modes/src/ecb.rs:16-50: same method order."""
    enc = re.findall(r"state\.(\w+)\(", py_fn(M.ecb_encrypt))
    assert enc == ["xor_key_enc", "sub_bytes", "shift_rows", "mix_columns", "xor_key_enc", "sub_bytes", "shift_rows", "xor_key_enc"]
    dec = re.findall(r"state\.(\w+)\(", py_fn(M.ecb_decrypt))
    assert dec == ["xor_key_enc", "inv_shift_rows", "inv_sub_bytes", "xor_key_enc", "inv_mix_columns", "inv_shift_rows", "inv_sub_bytes", "xor_key_enc"]


def test_xor_chain_rebalance_is_the_same_procedure_on_both_sides():
    """rebalance_xor_chains of hip.rs against the model's: the same absorption rule (xor / xnor, dead output slot, one use,
    by an xor / xnor), the same parity (xnor, negation bits 0 and 1), the earliest-two-first merge with (level, arrival, slot)
    ordering, spare slots popped for the inner gates, the root written to the original output with the parity in `neg`; and
    Engine::flush applies it under the same condition as ServerKey.flush of the model."""
    r = re.sub(r"\s+", "", rust_fn("hip.rs", "rebalance_xor_chains"))
    p = re.sub(r"\s+", "", py_fn(M.rebalance_xor_chains))
    # absorption rule, term by term and in the same order
    assert "is_xor(&gates[i])&&is_dead[gates[i].outasusize]&&uses[i]==1&&is_xor(&gates[user[i]])" in r
    assert 'is_xor(g)andg["out"]indeadanduses[i]==1andis_xor(gates[user[i]])' in p
    # parity
    assert "parity^=(x.op==sys::TFA_XNOR)asu32^(x.neg&1)^((x.neg>>1)&1)" in r
    assert 'parity^=int(x["op"]==XNOR)^(x["neg"]&1)^((x["neg"]>>1)&1)' in p
    # heap items (level, arrival, slot), inner gates on spare slots one level above the later operand, root on the gate's own output
    assert "heap.push(Reverse((level[sasusize],seq,s)))" in r and "heap.append((level.get(s,0),len(heap),s))" in p
    assert "(spare.pop().unwrap(),a.0.max(b.0)+1)" in r and "spare.pop(),max(a[0],b[0])+1" in p
    assert "sys::tfa_gate{op:sys::TFA_XOR,in0:a.2,in1:b.2,in2:0,out:slot,neg:0}" in r
    assert "dict(op=XOR,in0=a[2],in1=b[2],in2=0,out=slot,neg=0)" in p
    assert "sys::tfa_gate{op:sys::TFA_XOR,in0:a.2,in1:b.2,in2:0,out:g.out,neg:parity},a.0.max(b.0)+1" in r
    assert 'dict(op=XOR,in0=a[2],in1=b[2],in2=0,out=g["out"],neg=parity),max(a[0],b[0])+1' in p
    assert r.count("whileheap.len()>2") == 1 and p.count("whilelen(heap)>2") == 1
    # applied before the stages are submitted, only with pending gates, on the released list
    f = re.sub(r"\s+", "", rust_fn("hip.rs", "flush"))
    assert "ifself.xor_rebalance&&self.pending_gates>0{" in f and "rebalance_xor_chains(levels,&dead,self.next_freshasusize)" in f
    assert f.index("rebalance_xor_chains") < f.index("tfa_stage_submit")
    pf = re.sub(r"\s+", "", py_fn(M.ServerKey.flush))
    assert "ifself.xor_rebalanceandself.pending_gates:" in pf and "rebalance_xor_chains(levels,set(self.released))" in pf
