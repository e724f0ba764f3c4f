// boolfilter.cpp — plan-time compilation of a step's filter into atoms + a truth table (boolfilter.h).  Host C++.
#include "boolfilter.h"

#include <array>
#include <cstring>
#include <vector>

#include "expr.h"
#include "plan.h"

namespace mq {

namespace {

struct Sym {               // one value on the symbolic stack
  enum Kind { COL, LIT, BOOL, VAL } kind;   // VAL: an arithmetic value (only a program atom can consume it)
  int first = -1;          // the first node of the subtree that computes it (-1: not a contiguous run of this expression's nodes)
  bool casted = false;     // COL: under a cast (transparent for a range atom; a program atom restates the cast)
  int col = -1;            // COL: physical column
  int type = 0;            // COL / LIT: the value's integer type (after transparent casts)
  bool nullable = false;   // COL
  int64_t ival = 0;        // LIT
  std::vector<DevExprNode> prog;  // BOOL: reduced postfix program; atoms are placeholders (op EX_LIT, type INT8, arg = -1 - atom)
};

struct Compiler {
  const mi355q_plan& plan;     // the caller's plan (physical column descriptors)
  const DevExprSet& xs;        // its lowered expressions
  std::vector<BoolAtom> atoms;
  std::vector<int> atom_col;   // physical column of each atom
  bool ok = true;
  // program atoms (regprog.h): placeholder id kProgBase + k; the filter columns their operands read
  static constexpr int kProgBase = 1000;
  std::vector<RegProg> progs;
  std::vector<std::array<int, 2>> prog_ops;  // the physical columns behind each program's operand slots (-1: unused)

  // nodes [first, last] of expression k as a program atom (the BOOLEAN of a comparison / IS NULL over an arithmetic value,
  // two columns, a DOUBLE column, ...); -1: not a shape regprog.h states, or no room
  int add_prog_atom(const DevExpr& e, int first, int last) {
    if (first < 0) return -1;
    int ops[2] = {-1, -1};  // the physical columns behind operand slots 0 / 1
    auto slot_of = [&](int col) -> int {
      for (int i = 0; i < 2; ++i) {
        if (ops[i] == col) return i;
        if (ops[i] < 0) {
          ops[i] = col;
          return i;
        }
      }
      return -1;  // a third column: more than two live operands
    };
    RegProg rp;
    if (!rp_compile(e, first, last, xs.n_cols, slot_of, &rp)) return -1;
    if (rp.type != MI355Q_INT8) return -1;
    rp.n_ops = ops[1] >= 0 ? 2 : ops[0] >= 0 ? 1 : 0;
    for (size_t i = 0; i < progs.size(); ++i)
      if (std::memcmp(&progs[i], &rp, sizeof(RegProg)) == 0 && prog_ops[i][0] == ops[0] && prog_ops[i][1] == ops[1]) return kProgBase + (int)i;
    if ((int)progs.size() >= kBfMaxProgs) return -1;
    progs.push_back(rp);
    prog_ops.push_back({ops[0], ops[1]});
    return kProgBase + (int)progs.size() - 1;
  }
  bool prog_atom_sym(const DevExpr& e, int first, int last, Sym* r) {
    const int id = add_prog_atom(e, first, last);
    if (id < 0) return false;
    r->kind = Sym::BOOL;
    r->first = -1;
    r->prog.clear();
    r->prog.push_back(placeholder(id, progs[(size_t)(id - kProgBase)].nullable != 0));
    return true;
  }

  int add_atom(int col, const BoolAtom& a) {
    for (size_t i = 0; i < atoms.size(); ++i)
      if (atom_col[i] == col && atoms[i].lo == a.lo && atoms[i].hi == a.hi && atoms[i].negate == a.negate &&
          atoms[i].nullable == a.nullable && atoms[i].null_val == a.null_val)
        return (int)i;
    atoms.push_back(a);
    atom_col.push_back(col);
    return (int)atoms.size() - 1;
  }

  // `column <op> literal` at the comparison's operand type -> atom; false if the shape is not taken
  bool cmp_atom(const Sym& c, int q_op, int cmp_type, int64_t lit, int* atom_out) {
    const mi355q_col_desc& cd = plan.cols[c.col];
    const int code = col_type_code(cd);
    if (code != MI355Q_INT32 && code != MI355Q_INT64) return false;   // plain 4- / 8-byte integer chunks only
    if (cmp_type != MI355Q_INT32 && cmp_type != MI355Q_INT64) return false;
    DevQual dq{};
    dq.col = c.col;
    dq.op = q_op;
    dq.type = cmp_type;
    dq.nullable = cd.nullable != 0;
    dq.ival = lit;
    fast::RangeFilter f;
    if (!fast::make_range_filter(dq, &f)) return false;
    BoolAtom a{};
    a.lo = f.lo;
    a.hi = f.hi;
    a.negate = f.negate;
    // the NULL of the row is the COLUMN's pattern (a transparent cast maps it to the wider type's, a comparison with a
    // NULL operand is NULL): recognised before the range test, on the column's own value
    a.nullable = f.nullable;
    a.null_val = int_null_of(code);
    if (q_op == MI355Q_IS_NULL && cd.nullable) {  // (make_range_filter states it as the one-value range of the sentinel)
      a.lo = a.hi = a.null_val;
      a.nullable = 0;
    }
    *atom_out = add_atom(c.col, a);
    return true;
  }

  static DevExprNode placeholder(int atom, bool nullable) {
    DevExprNode n{};
    n.op = MI355Q_EX_LIT;
    n.type = MI355Q_INT8;
    n.arg = -1 - atom;
    n.flags = nullable ? EXF_NULLABLE : 0;
    return n;
  }

  // does a reduced program hold a program atom that can raise?
  bool holds_raising_atom(const std::vector<DevExprNode>& prog) const {
    for (const DevExprNode& n : prog)
      if (n.op == MI355Q_EX_LIT && n.arg < 0 && -1 - n.arg >= kProgBase && progs[(size_t)(-1 - n.arg - kProgBase)].can_raise) return true;
    return false;
  }

  bool walk(int k, Sym* result) {
    const DevExpr& e = xs.e[k];
    std::vector<Sym> st;
    for (int i = 0; i < e.n_nodes; ++i) {
      const DevExprNode& n = e.nodes[i];
      switch (n.op) {
        case MI355Q_EX_COL: {
          Sym s;
          if (n.arg >= xs.n_cols) {  // the value of an earlier expression: expanded in place.  Where a filter of comparisons is
                                     // evaluated does not matter — but an expression that can RAISE is also evaluated on its
                                     // own, for every row, ahead of the one that reads it: its errors are not this root's
            if (!walk(n.arg - xs.n_cols, &s) || s.kind != Sym::BOOL || holds_raising_atom(s.prog)) return false;
            s.first = -1;
          } else {
            if (!ex_is_int(n.type) && n.type != MI355Q_DOUBLE) return false;
            s.kind = Sym::COL;
            s.first = i;
            s.col = n.arg;
            s.type = n.type;
            s.nullable = (n.flags & EXF_NULLABLE) != 0;
          }
          st.push_back(std::move(s));
          break;
        }
        case MI355Q_EX_LIT: {
          if (n.arg != 0 || !(ex_is_int(n.type) || n.type == MI355Q_DOUBLE)) return false;  // (no NULL literal, no FLOAT)
          Sym s;
          s.kind = Sym::LIT;
          s.first = i;
          s.type = n.type;
          s.ival = n.ilit;
          st.push_back(std::move(s));
          break;
        }
        case MI355Q_EX_CAST: {
          if (st.empty()) return false;
          Sym& t = st.back();
          if (t.kind == Sym::COL && ex_is_int(n.type) && ex_is_int(t.type) && plain_width(n.type) >= plain_width(t.type)) {
            t.type = n.type;   // widening: transparent for a range atom
            t.casted = true;
          } else if (t.kind == Sym::LIT && ex_is_int(n.type) && ex_is_int(t.type) && t.ival <= ex_int_max(n.type) && t.ival > ex_int_min(n.type)) {
            t.type = n.type;
            t.casted = true;
          } else if (t.kind == Sym::COL || t.kind == Sym::LIT || t.kind == Sym::VAL) {
            t.kind = Sym::VAL;  // narrowing, to / from floating point, over an arithmetic value: a step of a program atom
            t.type = n.type;
          } else {
            return false;
          }
          break;
        }
        case MI355Q_EX_ADD: case MI355Q_EX_SUB: case MI355Q_EX_MUL: case MI355Q_EX_DIV: case MI355Q_EX_MOD: {
          if (st.size() < 2) return false;
          Sym b = std::move(st.back());
          st.pop_back();
          Sym& a = st.back();
          if (a.kind == Sym::BOOL || b.kind == Sym::BOOL || a.first < 0 || b.first < 0) return false;
          a.kind = Sym::VAL;
          a.type = n.type;
          break;
        }
        case MI355Q_EX_UMINUS: {
          if (st.empty() || st.back().kind == Sym::BOOL || st.back().first < 0) return false;
          st.back().kind = Sym::VAL;
          break;
        }
        case MI355Q_EX_EQ: case MI355Q_EX_NE: case MI355Q_EX_LT: case MI355Q_EX_LE: case MI355Q_EX_GT: case MI355Q_EX_GE: {
          if (st.size() < 2) return false;
          Sym b = std::move(st.back());
          st.pop_back();
          Sym a = std::move(st.back());
          st.pop_back();
          if (a.kind == Sym::BOOL || b.kind == Sym::BOOL) return false;
          int op = 0;
          const Sym* c = nullptr;
          int64_t lit = 0;
          if (a.kind == Sym::COL && b.kind == Sym::LIT) {
            c = &a;
            lit = b.ival;
            op = n.op == MI355Q_EX_EQ ? MI355Q_EQ : n.op == MI355Q_EX_NE ? MI355Q_NE : n.op == MI355Q_EX_LT ? MI355Q_LT
                 : n.op == MI355Q_EX_LE ? MI355Q_LE : n.op == MI355Q_EX_GT ? MI355Q_GT : MI355Q_GE;
          } else if (a.kind == Sym::LIT && b.kind == Sym::COL) {  // literal <op> column: the mirrored comparison
            c = &b;
            lit = a.ival;
            op = n.op == MI355Q_EX_EQ ? MI355Q_EQ : n.op == MI355Q_EX_NE ? MI355Q_NE : n.op == MI355Q_EX_LT ? MI355Q_GT
                 : n.op == MI355Q_EX_LE ? MI355Q_GE : n.op == MI355Q_EX_GT ? MI355Q_LT : MI355Q_LE;
          }
          Sym r;
          int atom;
          if (c && ex_is_int(c->type) && cmp_atom(*c, op, n.arg, lit, &atom)) {
            r.kind = Sym::BOOL;
            r.prog.push_back(placeholder(atom, atoms[atom].nullable != 0));
          } else if (!prog_atom_sym(e, a.first, i, &r)) {  // two columns, an arithmetic operand, a DOUBLE column: a program atom
            return false;
          }
          st.push_back(std::move(r));
          break;
        }
        case MI355Q_EX_IS_NULL: {
          if (st.empty()) return false;
          Sym a = std::move(st.back());
          st.pop_back();
          Sym r;
          r.kind = Sym::BOOL;
          int atom;
          if (a.kind == Sym::COL && ex_is_int(a.type) &&
              cmp_atom(a, MI355Q_IS_NULL, a.type == MI355Q_INT64 ? MI355Q_INT64 : MI355Q_INT32, 0, &atom)) {
            r.prog.push_back(placeholder(atom, false));
          } else if (a.kind == Sym::BOOL) {
            r.prog = std::move(a.prog);
            r.prog.push_back(n);
          } else if (!prog_atom_sym(e, a.first, i, &r)) {  // `(a + b) IS NULL`, a DOUBLE column
            return false;
          }
          st.push_back(std::move(r));
          break;
        }
        case MI355Q_EX_NOT: {
          if (st.empty() || st.back().kind != Sym::BOOL) return false;
          st.back().prog.push_back(n);
          break;
        }
        case MI355Q_EX_AND:
        case MI355Q_EX_OR: {
          if (st.size() < 2) return false;
          Sym b = std::move(st.back());
          st.pop_back();
          Sym& a = st.back();
          if (a.kind != Sym::BOOL || b.kind != Sym::BOOL) return false;
          a.prog.insert(a.prog.end(), b.prog.begin(), b.prog.end());
          a.prog.push_back(n);
          break;
        }
        default:
          return false;  // CASE: the projection pass
      }
    }
    if (st.size() != 1) return false;
    *result = std::move(st[0]);
    return true;
  }
};

// A reduced program (placeholders, NOT, AND / OR in both forms, IS NULL) for one state vector: the BOOLEAN it leaves and
// — *raiser — 0, or 1 + the PROGRAM atom whose error is the value's.  The loop of expr.h eval_expr over the same ex_*
// functions; the error a value carries is here the atom's tag instead of its code (ex_logic / ex_is_null only ask
// whether there is one and keep the first).
int64_t eval_reduced(const std::vector<DevExprNode>& prog, const int* state_of_placeholder /* by -1 - arg */, int n_range,
                     int32_t* raiser) {
  int64_t st[MI355Q_MAX_EXPR_NODES + 1] = {};
  int32_t es[MI355Q_MAX_EXPR_NODES + 1] = {};
  int sp = 0;
  for (const DevExprNode& n : prog) {
    switch (n.op) {
      case MI355Q_EX_LIT: {
        const int id = -1 - n.arg;
        const int pos = id >= Compiler::kProgBase ? n_range + (id - Compiler::kProgBase) : id;
        const int s = state_of_placeholder[pos];
        st[sp] = s == 1 ? 1 : s == 2 ? plain_int_null(MI355Q_INT8) : 0;
        es[sp] = s == 3 ? 1 + (id - Compiler::kProgBase) : 0;
        ++sp;
        break;
      }
      case MI355Q_EX_NOT: st[sp - 1] = ex_not(n, st[sp - 1]); break;
      case MI355Q_EX_IS_NULL: st[sp - 1] = ex_is_null(n, st[sp - 1], es[sp - 1]); break;
      default:  // MI355Q_EX_AND / _OR
        --sp;
        st[sp - 1] = ex_logic(n, st[sp - 1], st[sp], es[sp - 1], es[sp]);
    }
  }
  *raiser = es[0];
  return st[0];
}

}  // namespace

bool compile_bool_filter(const mi355q_plan& plan, BoolFilterHost* out, mi355q_plan* rest) {
  // (n_exprs = 0: a filter of plain quals alone — several of them in front of a family that filters on one column, api.cpp)
  if (plan.n_exprs < 0 || plan.n_exprs > MI355Q_MAX_EXPRS || plan.n_quals <= 0 || plan.n_quals > MI355Q_MAX_QUALS) return false;
  mi355q_plan lp;
  DevExprSet xs;
  if (lower_exprs(plan, &lp, &xs) != MI355Q_OK) return false;
  const int np = plan.n_cols;
  // nothing but the filter may read an expression
  if (expr_qual_mask(plan) != (1u << plan.n_exprs) - 1u) return false;
  for (int g = 0; g < plan.n_group_cols && g < MI355Q_MAX_GROUP_COLS; ++g)
    if (plan.group_cols[g] >= np) return false;
  for (int i = 0; i < plan.n_targets && i < MI355Q_MAX_TARGETS; ++i) {
    const mi355q_target& t = plan.targets[i];
    if (t.table == 0 && t.col >= np && t.agg != MI355Q_PROJECT_KEY) return false;
    if ((t.agg == MI355Q_COUNT_IF || t.agg == MI355Q_SUM_IF) && t.cond.col >= np) return false;
  }
  if (plan.join_outer_col >= np) return false;

  Compiler c{plan, xs, {}, {}, true, {}, {}};
  std::vector<int> plain_atoms;                    // conjuncts that are atoms themselves: must be TRUE
  struct Root {
    int expr;                                      // (-1: a NOT NULL test stated as a program)
    std::vector<DevExprNode> prog;
  };
  std::vector<Root> roots;                         // conjuncts that are BOOLEAN programs: must evaluate to 1
  for (int i = 0; i < plan.n_quals; ++i) {
    const mi355q_qual& q = plan.quals[i];
    if (MI355Q_QUAL_OR_GROUP(q.op) != 0 || q.op < 0 || (q.op >> 16) != 0) return false;
    if (q.col < 0 || q.col >= np + plan.n_exprs) return false;
    if (q.col < np) {
      Sym s;
      s.kind = Sym::COL;
      s.col = q.col;
      const int code = col_type_code(plan.cols[q.col]);
      if (code != MI355Q_INT32 && code != MI355Q_INT64) return false;
      s.type = code;
      int atom;
      const int op = MI355Q_QUAL_OP(q.op);
      if (op == MI355Q_IS_NOT_NULL) {  // NOT(IS NULL): the IS NULL atom under a NOT
        if (!c.cmp_atom(s, MI355Q_IS_NULL, code, 0, &atom)) return false;
        std::vector<DevExprNode> prog{Compiler::placeholder(atom, false)};
        DevExprNode nn{};
        nn.op = MI355Q_EX_NOT;
        nn.type = MI355Q_INT8;
        prog.push_back(nn);
        roots.push_back(Root{-1, std::move(prog)});
      } else {
        if (!c.cmp_atom(s, op, code, q.ival, &atom)) return false;
        plain_atoms.push_back(atom);
      }
    } else {
      // `BOOLEAN expression = 1`: the expression is TRUE (toBool; NULL is not)
      const DevExpr& e = xs.e[q.col - np];
      if (MI355Q_QUAL_OP(q.op) != MI355Q_EQ || q.ival != 1 || e.type != MI355Q_INT8) return false;
      Sym r;
      if (!c.walk(q.col - np, &r) || r.kind != Sym::BOOL) return false;
      if ((int)r.prog.size() > MI355Q_MAX_EXPR_NODES) return false;
      roots.push_back(Root{q.col - np, std::move(r.prog)});
    }
  }
  // an expression that can raise is evaluated for every row whether a qual names it or not (the row function evaluates the
  // filter's expressions in order before any qual): every expression must be a root here, each once
  {
    bool any_raise = false;
    for (const RegProg& rp : c.progs) any_raise = any_raise || rp.can_raise;
    if (any_raise) {
      uint32_t seen = 0;
      for (const Root& r : roots)
        if (r.expr >= 0) {
          if (seen & (1u << r.expr)) return false;
          seen |= 1u << r.expr;
        }
      if (seen != (1u << plan.n_exprs) - 1u) return false;
    }
  }
  const int na = (int)c.atoms.size(), npg = (int)c.progs.size();
  if (na + npg < 1 || na + npg > kBfMaxAtoms) return false;
  // ---- the filter's columns: those of the range atoms (grouped: the kernels walk them in a fixed order), then the ones
  // only programs read
  BoolFilterHost& o = *out;
  std::memset(&o, 0, sizeof(o));
  auto col_slot = [&](int col, bool add) -> int {
    for (int k = 0; k < o.bf.n_cols; ++k)
      if (o.bf.col[k] == col) return k;
    if (!add || o.bf.n_cols >= kBfMaxCols) return -1;
    o.bf.col[o.bf.n_cols] = col;
    o.bf.col_type[o.bf.n_cols] = col_type_code(plan.cols[col]);
    return o.bf.n_cols++;
  };
  for (int i = 0; i < na; ++i)
    if (col_slot(c.atom_col[i], true) < 0) return false;
  std::vector<int> order;   // new position -> old atom
  std::vector<int> new_of_old(na, -1);
  for (int k = 0; k < o.bf.n_cols; ++k)
    for (int i = 0; i < na; ++i)
      if (c.atom_col[i] == o.bf.col[k]) {
        new_of_old[i] = (int)order.size();
        o.bf.atom[order.size()] = c.atoms[i];
        order.push_back(i);
        ++o.bf.atoms_of_col[k];
      }
  o.bf.n_atoms = na;
  o.bf.n_progs = npg;
  for (int k = 0; k < npg; ++k) {
    for (int i = 0; i < 2; ++i) {
      const int col = c.prog_ops[(size_t)k][i];
      const int slot = col < 0 ? 0 : col_slot(col, true);
      if (slot < 0) return false;
      o.bf.prog_op[k][i] = slot;
    }
    o.bf.prog[k] = c.progs[(size_t)k];
    o.bf.any_raise |= c.progs[(size_t)k].can_raise;
  }
  o.bf.all_lean = 1;  // (vacuously so without programs: range atoms alone also take the lean member of the pre-pass)
  o.bf.all_i32 = 1;
  for (int k = 0; k < o.bf.n_cols; ++k) o.bf.all_i32 = o.bf.all_i32 && o.bf.col_type[k] == MI355Q_INT32;
  for (int k = 0; k < npg; ++k) {
    const int32_t types[2] = {o.bf.col_type[o.bf.prog_op[k][0]], c.prog_ops[(size_t)k][1] >= 0 ? o.bf.col_type[o.bf.prog_op[k][1]] : MI355Q_INT32};
    pair_atom_of(o.bf.prog[k], types, &o.bf.pair[k]);
    o.bf.all_lean = o.bf.all_lean && o.bf.pair[k].lean;
  }
  // ---- the truth table: the filter's own programs, run by the evaluator's functions, once per state vector
  int radix[kBfMaxAtoms];
  uint32_t n_states = 1;
  for (int i = 0; i < na + npg; ++i) {
    radix[i] = i >= na && o.bf.prog[i - na].can_raise ? 4 : 3;
    n_states *= (uint32_t)radix[i];
    if (n_states > 6561u) return false;
  }
  if (o.bf.any_raise && n_states > (uint32_t)kBfErrStates) return false;
  o.table_words = (int)((n_states + 31u) >> 5);
  for (const Root& r : roots) {  // (depth check: the evaluator's stack)
    int sp = 0, deepest = 0;
    for (const DevExprNode& n : r.prog) {
      if (n.op == MI355Q_EX_LIT) ++sp;
      else if (n.op == MI355Q_EX_AND || n.op == MI355Q_EX_OR) --sp;
      deepest = sp > deepest ? sp : deepest;
    }
    if (deepest > MI355Q_MAX_EXPR_STACK) return false;
  }
  // the order errors surface in: the expressions' own (a plain IS NOT NULL test cannot raise)
  std::vector<const Root*> by_expr;
  for (const Root& r : roots) by_expr.push_back(&r);
  for (size_t i = 1; i < by_expr.size(); ++i)
    for (size_t j = i; j > 0 && by_expr[j]->expr < by_expr[j - 1]->expr; --j) std::swap(by_expr[j], by_expr[j - 1]);
  int state[kBfMaxAtoms];       // by position in the state index (range atoms regrouped, then programs)
  int state_old[kBfMaxAtoms];   // as the placeholders name them: range atom ids, then programs
  for (uint32_t idx = 0; idx < n_states; ++idx) {
    uint32_t x = idx;
    bool possible = true;
    for (int i = 0; i < na + npg; ++i) {
      state[i] = (int)(x % (uint32_t)radix[i]);
      x /= (uint32_t)radix[i];
      const bool nullable = i < na ? o.bf.atom[i].nullable != 0 : o.bf.prog[i - na].nullable != 0;
      if (state[i] == 2 && !nullable) possible = false;
    }
    if (!possible) continue;
    for (int i = 0; i < na; ++i) state_old[i] = state[new_of_old[i]];
    for (int i = 0; i < npg; ++i) state_old[na + i] = state[na + i];
    bool pass = true;
    int32_t raiser = 0;
    for (int a : plain_atoms) pass = pass && state_old[a] == 1;
    for (const Root* r : by_expr) {
      int32_t tag = 0;
      const int64_t v = eval_reduced(r->prog, state_old, na, &tag);
      if (tag && !raiser) raiser = tag;
      pass = pass && v == 1;
    }
    if (raiser) {
      o.bf.etable[idx >> 3] |= (uint32_t)raiser << ((idx & 7u) * 4u);
      continue;  // (the row ends the step: it passes nothing)
    }
    if (pass) o.bf.table[idx >> 5] |= 1u << (idx & 31u);
  }
  // ---- what is left of the plan: no quals, no expressions
  *rest = plan;
  rest->n_quals = 0;
  rest->n_exprs = 0;
  return true;
}

}  // namespace mq
