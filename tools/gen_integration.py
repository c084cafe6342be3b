"""INTEGRATION.md section 1 from include/sandstorm_hip.h: the Rust `extern "C"` block a `hip` feature of ministark would hold, one
line per entry point of the header, argument for argument (the hand-written block had drifted - an in-place
ss_bitrev_permute32, ss_comm_all_gather's arguments in another order).  The block sits between two marker comments of
INTEGRATION.md; `python tools/gen_integration.py` rewrites it, `--check` fails if it is stale (tests/test_host_abi.py)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sandstorm_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED: tools/gen_integration.py -->", "<!-- END GENERATED -->"

SCALARS = {"uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int": "c_int", "size_t": "usize", "int64_t": "i64", "void": "c_void",
           "char": "c_char", "double": "f64", "float": "f32", "ss_status": "c_int", "ss_ctx": "SsCtx", "ss_comm": "SsComm", "ss_air_program": "SsAirProgram",
           "ss_perm_operand": "SsPermOperand", "ss_gather_job": "SsGatherJob", "uint16_t": "u16", "ss_trace_layout": "SsTraceLayout",
           "ss_trace_cell": "SsTraceCell", "ss_trace_rc_plan": "SsTraceRcPlan"}


def prototypes(text=None):
    """[(name, return C type, [(C type, name)])] in header order"""
    if text is None:
        with open(HEADER) as f:
            text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    out = []
    for stmt in text.split(";"):
        s = " ".join(stmt.split())
        m = re.match(r"^(?:extern \"C\" \{ )?(ss_status|void|const char \*|uint32_t) ?(ss_\w+) ?\((.*)\)$", s)
        if not m:
            continue
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"^(.*?)(\w+)(\[\d*\])?$", a)
                ctype, pname, arr = am.group(1).strip(), am.group(2), am.group(3)
                if arr:
                    ctype += " *" + ("/*%s*/" % arr)
                params.append((ctype, pname))
        out.append((name, ret, params))
    return out


def rust_type(ctype):
    note = ""
    m = re.search(r"/\*(\[\d*\])\*/", ctype)
    if m:
        note, ctype = " /* %s */" % m.group(1), ctype.replace(m.group(0), "")
    toks = ctype.replace("*", " * ").split()
    # parse left to right: [const] base, then a sequence of ('*', const?) declarators
    const_base = False
    i = 0
    if toks[i] == "const":
        const_base, i = True, i + 1
    base = SCALARS[toks[i]]
    i += 1
    if i < len(toks) and toks[i] == "const":           # "T const"
        const_base, i = True, i + 1
    ty, pointee_const = base, const_base
    while i < len(toks):
        assert toks[i] == "*", ctype
        ty = ("*const " if pointee_const else "*mut ") + ty
        i += 1
        pointee_const = False
        if i < len(toks) and toks[i] == "const":
            pointee_const, i = True, i + 1
    return ty + note


def rust_block():
    lines = ["#[repr(C)] pub struct SsCtx { _private: [u8; 0] }", "#[repr(C)] pub struct SsComm { _private: [u8; 0] }",
             "#[repr(C)] pub struct SsAirProgram {          // ss_air_program",
             "    pub code: *const u32, pub n_instr: u32,", "    pub consts: *const u64, pub n_consts: u32,",
             "    pub d_tables: *const u64, pub table_desc: *const u32, pub n_tables: u32,", "    pub n_slots: u32,", "}",
             "#[repr(C)] pub struct SsPermOperand {         // ss_perm_operand", "    pub d_data: *const u64, pub stride: u64, pub addr_offset: u64, pub value_offset: i64,", "}",
             "#[repr(C)] pub struct SsGatherJob {           // ss_gather_job",
             "    pub d_cols: *const *const c_void, pub ncols: u32, pub entry_bytes: u32, pub idx: *const u64, pub nidx: u32, pub out: *mut c_void,", "}",
             "#[repr(C)] pub struct SsTraceLayout { pub npc_pair: [u8; 8], pub rc_cell: [u8; 16], pub aux_cell: [u8; 16] }      // ss_trace_layout",
             "#[repr(C)] pub struct SsTraceCell { pub col: u32, pub off: u32, pub kind: u32, pub arg: u32 }                       // ss_trace_cell",
             "#[repr(C)] pub struct SsTraceRcPlan {         // ss_trace_rc_plan",
             "    pub n_slots: u64, pub n_given: u64, pub slot_rows: u64, pub addr_begin: u64, pub n_padding: u64, pub pad0: u64,",
             "    pub part_stride: u32, pub part_off: u32, pub pair_off: u32, pub rc_lo: u32, pub rc_hi: u32, pub ordered_step: u32, pub ordered_off: u32, pub unused_off: u32,", "}",
             "#[link(name = \"sandstorm_hip\")]", "extern \"C\" {"]
    for name, ret, params in prototypes():
        args = ", ".join("%s: %s" % (p if p not in ("in", "type", "ref", "mod") else p + "_", rust_type(t)) for t, p in params)
        r = "" if ret == "void" else " -> " + ("*const c_char" if ret == "const char *" else SCALARS[ret])
        lines.append("    pub fn %s(%s)%s;" % (name, args, r))
    lines.append("}")
    lines.append("fn check(code: c_int) { if code != 0 { panic!(\"{}\", unsafe { CStr::from_ptr(ss_last_error()) }.to_string_lossy()) } }")
    return "\n".join(lines)


def render(doc):
    i, j = doc.index(BEGIN), doc.index(END)
    return doc[:i] + BEGIN + "\n```rust\n" + rust_block() + "\n```\n" + doc[j:]


if __name__ == "__main__":
    with open(DOC) as f:
        doc = f.read()
    new = render(doc)
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else "INTEGRATION.md section 1 is stale: run python tools/gen_integration.py")
    with open(DOC, "w") as f:
        f.write(new)
    print("INTEGRATION.md: %d entry points" % len(prototypes()))
