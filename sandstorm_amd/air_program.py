"""Lowering of an AIR's composition constraint to the constraint-evaluation program
of the C ABI (ss_air_program, include/sandstorm_hip.h).

The reference describes constraints as an `Expr` DAG over the leaves X,
Constant, Trace(col, row_offset), Challenge, Hint and Periodic
(layouts/src/recursive/air.rs:61-1182), sums them with powers of one
composition coefficient and shares common nodes
(`composition_constraint` + `reuse_shared_nodes`, air.rs:1184-1200); ministark's
default AirConfig::eval_constraint then walks that DAG for every LDE point.  Here
the same DAG (hash-consed, so shared nodes are shared by construction) is lowered
once into straight-line code for a 4-accumulator, two-address register machine:

    word0 = opcode | dst << 8 | operand_kind << 12        word1 = operand payload

Challenges, hints and alpha^k are CONST operands; periodic columns and the
X^(n/k) zerofier inverses (both periodic in the LDE index) are TABLE operands.
"""


class OP:
    MOV, ADD, SUB, RSUB, MUL, INV, ST, OUT = range(8)


class SRC:
    ACC, SLOT, CONST, TRACE, TABLE, X = range(6)


def instr(op, dst, kind, payload):
    """-> [word0, word1]"""
    assert 0 <= dst < 4 and 0 <= payload < 2**32
    return [op | (dst << 8) | (kind << 12), payload]


def trace_payload(col, row_offset):
    assert 0 <= col < 256 and 0 <= row_offset < (1 << 24)
    return (col << 24) | row_offset


class Expr:
    """Hash-consed expression node.  kind in {x, const, const3, trace, table, add, sub, mul, inv}.
    `_skey` is a hash of the node's STRUCTURE (not of when it was built): commutative operands are ordered by it, so the DAG of
    an expression - and with it the lowered program, word for word - does not depend on what else the process built before
    (a compiled kernel recognises its program by the hash of the code words)."""
    __slots__ = ("kind", "args", "_id", "_skey")
    _pool = {}

    def __new__(cls, kind, *args):
        key = (kind,) + tuple(a._id if isinstance(a, Expr) else a for a in args)
        node = cls._pool.get(key)
        if node is None:
            import zlib
            node = object.__new__(cls)
            node.kind, node.args, node._id = kind, args, len(cls._pool)
            node._skey = zlib.crc32(repr((kind,) + tuple(("e", a._skey) if isinstance(a, Expr) else a for a in args)).encode())
            cls._pool[key] = node
        return node

    @property
    def is_leaf(self):
        return self.kind in ("x", "const", "const3", "sym", "trace", "table")

    def __add__(self, o): return Expr("add", *sorted((self, _wrap(o)), key=lambda e: (e._skey, e._id)))
    __radd__ = __add__
    def __sub__(self, o): return Expr("sub", self, _wrap(o))
    def __rsub__(self, o): return Expr("sub", _wrap(o), self)
    def __mul__(self, o): return Expr("mul", *sorted((self, _wrap(o)), key=lambda e: (e._skey, e._id)))
    __rmul__ = __mul__
    def __neg__(self): return Expr("sub", Const(0), self)
    def inverse(self): return Expr("inv", self)
    def __truediv__(self, o): return self * _wrap(o).inverse()

    def __pow__(self, k):
        assert isinstance(k, int) and k >= 1
        result, base = None, self
        while k:
            if k & 1:
                result = base if result is None else result * base
            k >>= 1
            if k:
                base = base * base
        return result


def _wrap(v):
    return v if isinstance(v, Expr) else Const(v)


X = Expr("x")


def Const(value):
    """A field constant (python int, canonical); challenges / hints / alpha^k are constants too."""
    return Expr("const", int(value))


def Const3(c0, c1, c2):
    """A constant of the cubic extension Fp[X]/(X^3 - 2) (the 64-bit field's challenges, hints, alpha^k): lower(.., ext=True)"""
    return Expr("const3", int(c0), int(c1), int(c2))


def Sym(name):
    """A constant known by NAME while the DAG is built (a challenge, a hint, alpha^k, a power of the trace generator): its value
    arrives with lower(..., symbols=) / evaluate_ext(..., symbols=).  Two statements of a layout then share one DAG, hence one
    program word for word - only the constant table differs - which is what lets a compiled kernel recognise the program."""
    return Expr("sym", str(name))


def Trace(col, row_offset=0):
    return Expr("trace", int(col), int(row_offset))


def Table(index):
    return Expr("table", int(index))


class Program:
    """code (list of u32 words), consts (python ints, canonical), n_slots"""

    def __init__(self):
        self.code, self.consts, self.n_slots = [], [], 0
        self._const_ix = {}

    @property
    def n_instr(self):
        return len(self.code) // 2

    def const_index(self, value, key=None):
        """key: what the constant is interned by (its value unless it is a named one)"""
        key = value if key is None else key
        ix = self._const_ix.get(key)
        if ix is None:
            ix = self._const_ix[key] = len(self.consts)
            self.consts.append(value)
        return ix


def lower(root, modulus, ext=False, symbols=None):
    """Expr DAG -> Program whose last instruction OUTs the value of `root`.
    ext: the program of the cubic-extension machine (ss_eval_quotient_gl64x3): every constant is a triple (an int v is (v, 0, 0)).
    symbols: {name: value} for the Sym leaves (interned by name: equal values of different names stay different constants).

    Tree-walk code generation with accumulator `dst` as the working register:
    operands that are leaves (or shared nodes already parked in a slot) are used in
    place; when both children need code the right one goes to acc[dst+1] (or, past
    the fourth accumulator, the left one is parked in a scratch slot).  A node with
    several parents is computed once and kept in a slot until its last use."""
    import sys
    sys.setrecursionlimit(max(100000, sys.getrecursionlimit()))
    prog = Program()
    uses, seen, stack = {}, set(), [root]
    while stack:
        n = stack.pop()
        if n._id in seen:
            continue
        seen.add(n._id)
        for a in n.args:
            if isinstance(a, Expr):
                uses[a._id] = uses.get(a._id, 0) + 1
                stack.append(a)
    uses[root._id] = uses.get(root._id, 0) + 1
    slot_of, free_slots = {}, []

    def alloc_slot():
        if free_slots:
            return free_slots.pop()
        prog.n_slots += 1
        return prog.n_slots - 1

    def operand(n):
        """(kind, payload) when n is usable in place, else None"""
        if n.kind == "x":
            return SRC.X, 0
        if n.kind == "const":
            return SRC.CONST, prog.const_index((n.args[0] % modulus, 0, 0) if ext else n.args[0] % modulus)
        if n.kind == "const3":
            assert ext, "extension-field constant in a base-field program"
            return SRC.CONST, prog.const_index(tuple(v % modulus for v in n.args))
        if n.kind == "sym":
            v = symbols[n.args[0]]
            v = tuple(c % modulus for c in v) if isinstance(v, tuple) else ((v % modulus, 0, 0) if ext else v % modulus)
            return SRC.CONST, prog.const_index(v, key=("sym", n.args[0]))
        if n.kind == "trace":
            return SRC.TRACE, trace_payload(*n.args)
        if n.kind == "table":
            return SRC.TABLE, n.args[0]
        if n._id in slot_of:
            return SRC.SLOT, slot_of[n._id]
        return None

    def consume(n):
        """one parent reference of n has been served"""
        if n.is_leaf:
            return
        uses[n._id] -= 1
        if uses[n._id] == 0 and n._id in slot_of:
            free_slots.append(slot_of.pop(n._id))

    def emit(op, dst, kind=0, payload=0):
        prog.code += instr(op, dst, kind, payload)

    need_of = {}

    def need(n):
        """accumulators a fresh evaluation of n takes (leaves: 0)"""
        if n.is_leaf:
            return 0
        v = need_of.get(n._id)
        if v is None:
            if n.kind == "inv":
                v = max(1, need(n.args[0]))
            else:
                a, b = need(n.args[0]), need(n.args[1])
                v = max(1, a if n.args[0] is n.args[1] else (a + 1 if a == b else max(a, b)))
            need_of[n._id] = v
        return v

    def gen(n, dst):
        """code leaving n in acc[dst]; serves one parent reference of n"""
        opnd = operand(n)
        if opnd is not None:
            emit(OP.MOV, dst, *opnd)
            consume(n)
            return
        if n.kind == "inv":
            gen(n.args[0], dst)
            emit(OP.INV, dst)
        else:
            l, r = n.args
            opc = {"add": OP.ADD, "sub": OP.SUB, "mul": OP.MUL}[n.kind]
            if l is r and operand(l) is None:
                uses[l._id] -= 1                      # both references served by one evaluation
                gen(l, dst)
                emit(opc, dst, SRC.ACC, dst)
            elif operand(r) is not None:
                gen(l, dst)
                emit(opc, dst, *operand(r))
                consume(r)
            elif operand(l) is not None:
                gen(r, dst)
                emit(OP.RSUB if n.kind == "sub" else opc, dst, *operand(l))
                consume(l)
            else:
                # both operands need code: the one that needs more accumulators goes first (Sethi-Ullman), whatever the
                # order the DAG holds them in; a subtraction the other way round is an RSUB
                first, second, swapped = (l, r, False) if need(l) >= need(r) else (r, l, True)
                rev = OP.RSUB if n.kind == "sub" else opc
                op_ab = rev if swapped else opc            # acc[dst] = first (op) second, in the DAG's sense
                gen(first, dst)
                if operand(second) is not None:        # it was a shared sub-node of the first
                    emit(op_ab, dst, *operand(second))
                    consume(second)
                elif dst + 1 < 4:
                    gen(second, dst + 1)
                    emit(op_ab, dst, SRC.ACC, dst + 1)
                else:
                    s = alloc_slot()
                    emit(OP.ST, dst, 0, s)
                    gen(second, dst)
                    emit((opc if swapped else rev), dst, SRC.SLOT, s)      # acc[dst] holds `second` now, the slot `first`
                    free_slots.append(s)
        uses[n._id] -= 1
        if uses[n._id] > 0:                            # more parents: park it
            slot_of[n._id] = alloc_slot()
            emit(OP.ST, dst, 0, slot_of[n._id])

    gen(root, 0)
    emit(OP.OUT, 0)
    return prog


def evaluate(root, modulus, x, trace_at, table_at):
    """Direct big-integer evaluation of the DAG at one point (the definition the
    lowered program must reproduce).  trace_at(col, off) / table_at(idx) -> int."""
    memo = {}

    def ev(n):
        v = memo.get(n._id)
        if v is not None:
            return v
        k = n.kind
        if k == "x":
            v = x
        elif k == "const":
            v = n.args[0] % modulus
        elif k == "trace":
            v = trace_at(*n.args)
        elif k == "table":
            v = table_at(n.args[0])
        elif k == "inv":
            a = ev(n.args[0])
            v = pow(a, -1, modulus) if a else 0
        else:
            a, b = ev(n.args[0]), ev(n.args[1])
            v = (a + b) % modulus if k == "add" else (a - b) % modulus if k == "sub" else a * b % modulus
        memo[n._id] = v
        return v

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    return ev(root)


def evaluate_ext(root, modulus, x, trace_at, table_at, nonresidue=2, symbols=None):
    """evaluate() over the cubic extension Fp[X]/(X^3 - nonresidue): x, trace_at(col, off), table_at(idx) -> triples; the
    out-of-domain side of the 64-bit field's AIR identity."""
    p = modulus

    def mul(a, b):
        d = [0] * 5
        for i in range(3):
            for j in range(3):
                d[i + j] += a[i] * b[j]
        return ((d[0] + nonresidue * d[3]) % p, (d[1] + nonresidue * d[4]) % p, d[2] % p)

    def inv(a):
        if not any(a):
            return (0, 0, 0)
        r, base, e = (1, 0, 0), a, p ** 3 - 2
        while e:
            if e & 1:
                r = mul(r, base)
            base, e = mul(base, base), e >> 1
        return r
    memo = {}

    def ev(n):
        v = memo.get(n._id)
        if v is not None:
            return v
        k = n.kind
        if k == "x":
            v = tuple(x)
        elif k == "const":
            v = (n.args[0] % p, 0, 0)
        elif k == "const3":
            v = tuple(c % p for c in n.args)
        elif k == "sym":
            sv = symbols[n.args[0]]
            v = tuple(c % p for c in sv) if isinstance(sv, tuple) else (sv % p, 0, 0)
        elif k == "trace":
            v = tuple(trace_at(*n.args))
        elif k == "table":
            v = tuple(table_at(n.args[0]))
        elif k == "inv":
            v = inv(ev(n.args[0]))
        else:
            a, b = ev(n.args[0]), ev(n.args[1])
            v = (tuple((s + t) % p for s, t in zip(a, b)) if k == "add" else tuple((s - t) % p for s, t in zip(a, b)) if k == "sub" else mul(a, b))
        memo[n._id] = v
        return v

    import sys
    sys.setrecursionlimit(max(100000, sys.getrecursionlimit()))
    return ev(root)
