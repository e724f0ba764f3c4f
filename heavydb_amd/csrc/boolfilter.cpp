// boolfilter.cpp — plan-time compilation of a step's filter into atoms + a truth table (boolfilter.h).  Host C++.
#include "boolfilter.h"

#include <cstring>
#include <vector>

#include "expr.h"
#include "plan.h"

namespace mq {

namespace {

struct Sym {               // one value on the symbolic stack
  enum Kind { COL, LIT, BOOL } kind;
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

  bool walk(int k, Sym* result) {
    const DevExpr& e = xs.e[k];
    std::vector<Sym> st;
    for (int i = 0; i < e.n_nodes; ++i) {
      const DevExprNode& n = e.nodes[i];
      switch (n.op) {
        case MI355Q_EX_COL: {
          Sym s;
          if (n.arg >= xs.n_cols) {  // the value of an earlier expression: expanded in place (a filter of comparisons cannot
                                     // raise, so where it is evaluated does not matter)
            if (!walk(n.arg - xs.n_cols, &s) || s.kind != Sym::BOOL) return false;
          } else {
            if (!ex_is_int(n.type)) return false;
            s.kind = Sym::COL;
            s.col = n.arg;
            s.type = n.type;
            s.nullable = (n.flags & EXF_NULLABLE) != 0;
          }
          st.push_back(std::move(s));
          break;
        }
        case MI355Q_EX_LIT: {
          if (n.arg != 0 || !ex_is_int(n.type)) return false;  // (no NULL literal, no floating point)
          Sym s;
          s.kind = Sym::LIT;
          s.type = n.type;
          s.ival = n.ilit;
          st.push_back(std::move(s));
          break;
        }
        case MI355Q_EX_CAST: {
          if (st.empty() || !ex_is_int(n.type)) return false;
          Sym& t = st.back();
          if (t.kind == Sym::COL) {
            if (plain_width(n.type) < plain_width(t.type)) return false;  // narrowing: can raise error 7
            t.type = n.type;
          } else if (t.kind == Sym::LIT) {
            if (t.ival > ex_int_max(n.type) || t.ival <= ex_int_min(n.type)) return false;
            t.type = n.type;
          } else {
            return false;
          }
          break;
        }
        case MI355Q_EX_EQ: case MI355Q_EX_NE: case MI355Q_EX_LT: case MI355Q_EX_LE: case MI355Q_EX_GT: case MI355Q_EX_GE: {
          if (st.size() < 2) return false;
          Sym b = std::move(st.back());
          st.pop_back();
          Sym a = std::move(st.back());
          st.pop_back();
          int op;
          const Sym* c;
          int64_t lit;
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
          } else {
            return false;
          }
          int atom;
          if (!cmp_atom(*c, op, n.arg, lit, &atom)) return false;
          Sym r;
          r.kind = Sym::BOOL;
          r.prog.push_back(placeholder(atom, atoms[atom].nullable != 0));
          st.push_back(std::move(r));
          break;
        }
        case MI355Q_EX_IS_NULL: {
          if (st.empty()) return false;
          Sym a = std::move(st.back());
          st.pop_back();
          Sym r;
          r.kind = Sym::BOOL;
          if (a.kind == Sym::COL) {
            int atom;
            if (!cmp_atom(a, MI355Q_IS_NULL, a.type == MI355Q_INT64 ? MI355Q_INT64 : MI355Q_INT32, 0, &atom)) return false;
            r.prog.push_back(placeholder(atom, false));
          } else if (a.kind == Sym::BOOL) {
            r.prog = std::move(a.prog);
            r.prog.push_back(n);
          } else {
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
          return false;  // arithmetic, CASE, unary minus: the projection pass
      }
    }
    if (st.size() != 1) return false;
    *result = std::move(st[0]);
    return true;
  }
};

}  // namespace

bool compile_bool_filter(const mi355q_plan& plan, BoolFilterHost* out, mi355q_plan* rest) {
  if (plan.n_exprs <= 0 || plan.n_exprs > MI355Q_MAX_EXPRS || plan.n_quals <= 0 || plan.n_quals > MI355Q_MAX_QUALS) return false;
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

  Compiler c{plan, xs, {}, {}, true};
  std::vector<int> plain_atoms;                    // conjuncts that are atoms themselves: must be TRUE
  std::vector<std::vector<DevExprNode>> roots;     // conjuncts that are BOOLEAN programs: must evaluate to 1
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
        roots.push_back(std::move(prog));
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
      roots.push_back(std::move(r.prog));
    }
  }
  const int na = (int)c.atoms.size();
  if (na < 1 || na > kBfMaxAtoms) return false;
  // ---- atoms grouped by column (the kernels walk their filter columns in a fixed order)
  BoolFilterHost& o = *out;
  std::memset(&o, 0, sizeof(o));
  std::vector<int> order;   // new position -> old atom
  for (int i = 0; i < na; ++i) {
    int slot = -1;
    for (int k = 0; k < o.bf.n_cols; ++k)
      if (o.bf.col[k] == c.atom_col[i]) slot = k;
    if (slot < 0) {
      if (o.bf.n_cols >= kBfMaxCols) return false;
      o.bf.col[o.bf.n_cols] = c.atom_col[i];
      o.bf.col_type[o.bf.n_cols] = col_type_code(plan.cols[c.atom_col[i]]);
      ++o.bf.n_cols;
    }
  }
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
  // ---- the truth table: the filter's own programs, run by the evaluator of the interpreter pass, once per state vector
  uint32_t n_states = 1;
  for (int i = 0; i < na; ++i) n_states *= 3u;
  o.table_words = (int)((n_states + 31u) >> 5);
  std::vector<DevExpr> progs(roots.size());
  for (size_t r = 0; r < roots.size(); ++r) {
    std::memset(&progs[r], 0, sizeof(DevExpr));
    progs[r].n_nodes = (int)roots[r].size();
    progs[r].type = MI355Q_INT8;
    for (size_t i = 0; i < roots[r].size(); ++i) progs[r].nodes[i] = roots[r][i];
    // (depth check: the evaluator's stack)
    int sp = 0, deepest = 0;
    for (const DevExprNode& n : roots[r]) {
      if (n.op == MI355Q_EX_LIT) ++sp;
      else if (n.op == MI355Q_EX_AND || n.op == MI355Q_EX_OR) --sp;
      deepest = sp > deepest ? sp : deepest;
    }
    if (deepest > MI355Q_MAX_EXPR_STACK) return false;
  }
  int state[kBfMaxAtoms];
  for (uint32_t idx = 0; idx < n_states; ++idx) {
    uint32_t x = idx;
    bool possible = true;
    for (int i = 0; i < na; ++i) {
      state[i] = (int)(x % 3u);
      x /= 3u;
      if (state[i] == 2 && !o.bf.atom[i].nullable) possible = false;
    }
    if (!possible) continue;
    bool pass = true;
    for (int a : plain_atoms) pass = pass && state[new_of_old[a]] == 1;
    for (size_t r = 0; r < progs.size() && pass; ++r) {
      DevExpr e = progs[r];
      for (int i = 0; i < e.n_nodes; ++i) {
        DevExprNode& n = e.nodes[i];
        if (n.op != MI355Q_EX_LIT || n.arg >= 0) continue;
        const int s = state[new_of_old[-1 - n.arg]];
        n.arg = s == 2 ? 1 : 0;                            // (1: "the NULL literal, pattern in ilit")
        n.ilit = s == 2 ? plain_int_null(MI355Q_INT8) : s;
        n.flit = 0.0;
      }
      int32_t err = 0;
      const int64_t v = eval_expr(e, nullptr, 0, &err);
      if (err) return false;  // (cannot happen: no node of the reduced program can raise)
      pass = v == 1;
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
