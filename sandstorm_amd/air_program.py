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
    """Hash-consed expression node.  kind in {x, const, trace, table, add, sub, mul, inv}."""
    __slots__ = ("kind", "args", "_id")
    _pool = {}

    def __new__(cls, kind, *args):
        key = (kind,) + tuple(a._id if isinstance(a, Expr) else a for a in args)
        node = cls._pool.get(key)
        if node is None:
            node = object.__new__(cls)
            node.kind, node.args, node._id = kind, args, len(cls._pool)
            cls._pool[key] = node
        return node

    @property
    def is_leaf(self):
        return self.kind in ("x", "const", "const3", "trace", "table")

    def __add__(self, o): return Expr("add", *sorted((self, _wrap(o)), key=lambda e: e._id))
    __radd__ = __add__
    def __sub__(self, o): return Expr("sub", self, _wrap(o))
    def __rsub__(self, o): return Expr("sub", _wrap(o), self)
    def __mul__(self, o): return Expr("mul", *sorted((self, _wrap(o)), key=lambda e: e._id))
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

    def const_index(self, value):
        ix = self._const_ix.get(value)
        if ix is None:
            ix = self._const_ix[value] = len(self.consts)
            self.consts.append(value)
        return ix


def lower(root, modulus, ext=False):
    """Expr DAG -> Program whose last instruction OUTs the value of `root`.
    ext: the program of the cubic-extension machine (ss_eval_quotient_gl64x3): every constant is a triple (an int v is (v, 0, 0)).

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
                gen(l, dst)
                if operand(r) is not None:             # r was a shared sub-node of l
                    emit(opc, dst, *operand(r))
                    consume(r)
                elif dst + 1 < 4:
                    gen(r, dst + 1)
                    emit(opc, dst, SRC.ACC, dst + 1)
                else:
                    s = alloc_slot()
                    emit(OP.ST, dst, 0, s)
                    gen(r, dst)
                    emit(OP.RSUB if n.kind == "sub" else opc, dst, SRC.SLOT, s)
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


def evaluate_ext(root, modulus, x, trace_at, table_at, nonresidue=2):
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
