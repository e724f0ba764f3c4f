/*
 * mi355q.h — C-ABI of the MI355X-native query-step executor.
 *
 * This is the drop-in boundary for ONE HeavyDB query step (scan/filter ->
 * hash group-by + aggregate -> hash-join probe).  Every entry point takes
 * plain pointers and sizes (no C++ / torch types) and returns a HeavyDB
 * error code (QueryEngine/enums.h:30-51): 0 = ok, >0 = persistent error,
 * <0 = ran out of group slots (caller resizes the table and retries, as
 * RelAlgExecutor.cpp:4143-4145, :4194-4231 does).
 *
 * Reference interfaces each entry point replaces (paths relative to the
 * heavyai/heavydb tree):
 *
 *   mi355q_qmd_init        GroupByAndAggregate::initQueryMemoryDescriptor
 *                          (QueryEngine/GroupByAndAggregate.cpp:859) =
 *                          getColRangeInfo (:232-365) + get_keyless_info (:489-648)
 *                          + QueryMemoryDescriptor::init
 *                          (Descriptors/QueryMemoryDescriptor.cpp:240-446)
 *   mi355q_execute         Executor::executeWorkUnit (QueryEngine/Execute.h:719,
 *                          Execute.cpp:2144) at the point ExecutionKernel::runImpl
 *                          has fetched chunks (ExecutionKernel.cpp:270-292, the seam
 *                          run_query_external uses, ExternalExecutor.h:66-69):
 *                          raw column pointers in, ResultSetStorage buffer out.
 *   mi355q_result_reduce   ResultSetStorage::reduce (ResultSetReduction.cpp:203)
 *   mi355q_result_*        ResultSet accessors (ResultSet.h:263 getNextRow,
 *                          :327 rowCount; ResultSetIteration.cpp:2457 isEmptyEntry;
 *                          ResultSetBufferAccessors.h:197 pair_to_double)
 *   mi355q_result_topk     ResultSet::sort (ResultSet.h:278) -> baselineSort /
 *                          TopKSort.cu for one order entry with a row limit
 *   mi355q_join_build      HashJoin::getInstance (JoinHashTable/HashJoin.cpp:286):
 *                          PerfectJoinHashTable (PerfectJoinHashTable.cpp:168) then
 *                          BaselineJoinHashTable (BaselineJoinHashTable.cpp:484)
 *   mi355q_shard_*         the multi-device merge of
 *                          Executor::reduceMultiDeviceResultSets (Execute.cpp:1772)
 *                          done on-device, for one-process-per-GPU operation.
 */
#ifndef MI355Q_H
#define MI355Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355Q_ABI_VERSION 7

#define MI355Q_MAX_COLS 16
#define MI355Q_MAX_QUALS 8
#define MI355Q_MAX_TARGETS 8
#define MI355Q_MAX_SLOTS 16
#define MI355Q_MAX_GROUP_COLS 4
#define MI355Q_MAX_EXPRS 8
#define MI355Q_MAX_EXPR_NODES 24
#define MI355Q_MAX_EXPR_STACK 8 /* values on the evaluation stack of a postfix program (CASE WHEN a >= 6 AND a <= 7 ... needs 5) */

/* ---- error codes: numeric values of heavyai::ErrorCode (enums.h:30-51) ---- */
#define MI355Q_OK 0
#define MI355Q_ERR_DIV_BY_ZERO 1
#define MI355Q_ERR_OUT_OF_GPU_MEM 2
#define MI355Q_ERR_OUT_OF_SLOTS 3
#define MI355Q_ERR_OUT_OF_CPU_MEM 6
#define MI355Q_ERR_OVERFLOW_OR_UNDERFLOW 7
#define MI355Q_ERR_OUT_OF_TIME 8
#define MI355Q_ERR_INTERRUPTED 9
/* library-level codes (outside the reference's range) */
#define MI355Q_ERR_INVALID_PLAN 100
#define MI355Q_ERR_UNSUPPORTED 101
#define MI355Q_ERR_HIP 102
#define MI355Q_ERR_JOIN_NOT_ONE_TO_ONE 103 /* reference: fill returns -1 -> 1:N rebuild */
#define MI355Q_ERR_JOIN_TABLE_FULL 104     /* reference: write_baseline_hash_slot -2 */
/* mi355q_wait only, and not an error: the result is complete and correct, but the step had to be RE-RUN inside
 * mi355q_wait (the partitioned family ran out of spill space and the direct member took over), i.e. after
 * mi355q_execute_async had returned.  Whatever the caller enqueued behind the first launches on the same stream
 * (mi355q_shard_pads, slices handed to a collective, a merge) read the table of the abandoned attempt and has to
 * be redone from the result as it is now. */
#define MI355Q_STEP_RECOMPUTED 110

/* ---- column types (fixed-width, as ColumnFetcher hands them over) ---- */
typedef enum mi355q_type {
  MI355Q_INT8 = 1,
  MI355Q_INT16 = 2,
  MI355Q_INT32 = 3,
  MI355Q_INT64 = 4,
  MI355Q_DOUBLE = 5,
  MI355Q_FLOAT = 6 /* kFLOAT, 4-byte chunks (fixed_width_float_decode, DecodersImpl.h:109-119).
                      Aggregates with a FLOAT argument work in single precision on the LOW 4
                      bytes of their 8-byte slot (takes_float_argument, Shared/TargetInfo.h:106;
                      agg_sum_float / agg_min_float / agg_max_float, RuntimeFunctions.cpp:1496-
                      1520); filter and aggregate argument only — not a group or join key */
} mi355q_type;

/* Column encodings ColumnFetcher hands over undecoded (EncodingType, Shared/sqltypes.h; the
 * decoders are chosen by get_col_decoder, QueryEngine/ColumnIR.cpp, and live in
 * DecodersImpl.h).  `type` of the column is always the STORAGE type of the chunk. */
typedef enum mi355q_encoding {
  MI355Q_ENC_NONE = 0,
  MI355Q_ENC_FIXED = 1, /* kENCODING_FIXED: integer stored narrower than its SQL type
                           (`logical_type`); NULL is the storage type's sentinel and is widened
                           to the logical sentinel on load (codgenAdjustFixedEncNull,
                           ColumnIR.cpp; fixed_width_int_decode DecodersImpl.h:27-55) */
  MI355Q_ENC_DICT = 2,  /* kENCODING_DICT string ids: 1/2-byte chunks are UNSIGNED
                           (fixed_width_unsigned_decode DecodersImpl.h:57-85), NULL = 255 / 65535,
                           widened to the INT32 sentinel; 4-byte chunks are plain int32 */
  MI355Q_ENC_DATE_IN_DAYS = 3 /* kENCODING_DATE_IN_DAYS: int16/int32 days since epoch ->
                                 int64 seconds (x 86400), NULL -> NULL_BIGINT
                                 (fixed_width_small_date_decode DecodersImpl.h:130-139) */
} mi355q_encoding;

/* comparison operators: numeric values of SQLOps (Shared/sqldefs.h:31-38) */
typedef enum mi355q_op {
  MI355Q_EQ = 0,
  MI355Q_NE = 2,
  MI355Q_LT = 3,
  MI355Q_GT = 4,
  MI355Q_LE = 5,
  MI355Q_GE = 6,
  /* unary quals on a column, no literal (kISNULL = 16, kISNOTNULL = 17).  `x IS NULL` is FALSE on a
   * NOT NULL column and `value == the type's inline NULL` otherwise (CodeGenerator::codegenIsNull,
   * LogicalIR.cpp:381-432; doubles compare with FCMP_OEQ against NULL_DOUBLE); `x IS NOT NULL` is its
   * negation (RelAlgTranslator.cpp:640-643 builds NOT(ISNULL(x))).  A qual `x IS NOT NULL` also makes
   * the GROUPED aggregates over x NOT NULL for this step (constrained_not_null,
   * OutputBufferInitialization.cpp:287,301-324 — init values; TargetExprBuilder.cpp:690 — plain
   * instead of _skip_val aggregates; GroupByAndAggregate.cpp:531 — the keyless rule of SUM). */
  MI355Q_IS_NULL = 16,
  MI355Q_IS_NOT_NULL = 17
} mi355q_op;

/* aggregates: numeric values of SQLAgg (Shared/sqldefs.h:76-90); PROJECT is a
 * non-aggregate target that projects the group key (TargetInfo.is_agg == false,
 * written with agg_id, TargetExprBuilder.cpp:58-72). */
typedef enum mi355q_agg {
  MI355Q_AVG = 0,
  MI355Q_MIN = 1,
  MI355Q_MAX = 2,
  MI355Q_SUM = 3,
  MI355Q_COUNT = 4,
  MI355Q_COUNT_IF = 10, /* COUNT_IF(cond): rows whose condition is TRUE (agg_count_if[_skip_val],
                           RuntimeFunctions.cpp:1356-1375) */
  MI355Q_SUM_IF = 11,   /* SUM_IF(col, cond): SUM over the rows whose condition is TRUE
                           (agg_sum_if*, RuntimeFunctions.cpp:1157-1161,1341-1346,1450-1456;
                           codegenConditionalAggregateCondValSelector: TRUE means == 1, a NULL
                           condition is not TRUE) */
  MI355Q_PROJECT_KEY = 100,
  /* a non-aggregate target of a PROJECTION step (no GROUP BY, no aggregate: `SELECT a, b + 1 FROM t WHERE ...`): the value
   * of outer column / expression `col` of every row that passes the quals, written with agg_id
   * (TargetExprCodegen::codegenAggregate, TargetExprBuilder.cpp:330-560; is_agg == false).  A step is a Projection when
   * n_group_cols == 0 and EVERY target is MI355Q_PROJECT.  Through a join (`table` = 1 reads the inner side): every JOINED
   * row is an entry — one per outer row over a one-to-one table, the whole matching set over a one-to-many table
   * (HashJoin::codegenMatchingSet, HashJoin.cpp:209; the entries of one outer row in the payload run's order, which
   * depends on the build order as in the reference); a LEFT join keeps an unmatched row once with the inner columns'
   * NULLs.  MI355Q_ERR_UNSUPPORTED: a one-to-many table together with projected expressions. */
  MI355Q_PROJECT = 101
} mi355q_agg;

/* join kinds: INNER drops outer rows without a match; LEFT keeps them once with every inner
 * column NULL (JoinType::LEFT; Executor::buildJoinLoops, IRCodegen.cpp, JoinLoop kinds
 * UpperBound/Set/Singleton with an outer-join "found" flag) */
typedef enum mi355q_join_kind { MI355Q_JOIN_INNER = 0, MI355Q_JOIN_LEFT = 1 } mi355q_join_kind;

/* QueryDescriptionType (enums.h:53-59) */
typedef enum mi355q_desc_type {
  MI355Q_GROUP_BY_PERFECT_HASH = 0,
  MI355Q_GROUP_BY_BASELINE_HASH = 1,
  /* QueryDescriptionType::Projection: one output entry per row that passes the quals.  Entry e holds
   * [ row's offset in its fragment (int64, the "key": get_scan_output_slot / get_columnar_scan_output_offset,
   * GroupByRuntime.cpp:242-269) | one slot per target ]; entries [row count, entry_count) keep the EMPTY_KEY_64 key of
   * an initialised buffer.  entry_count = scan_limit, or max_groups_buffer_entry_guess when the plan has none
   * (QueryMemoryDescriptor.cpp:394-410).  Row-wise: 8-byte slots (integers sign-extended, FLOAT widened to double);
   * columnar (output_columnar_hint): slot columns of the targets' LOGICAL widths (isLogicalSizedColumnsAllowed, :1129),
   * each align_to_int64(width * entry_count) bytes.  The entries come out in (fragment, row) order — the order the
   * reference's CPU executor produces with one kernel per fragment (its GPU kernel's order is that of an atomic
   * counter). */
  MI355Q_PROJECTION = 2,
  MI355Q_NON_GROUPED_AGGREGATE = 4
} mi355q_desc_type;

typedef struct mi355q_col_desc {
  int32_t type;     /* mi355q_type of the chunk as stored */
  int32_t nullable; /* 0 = NOT NULL; else NULL is the inline sentinel
                       (Shared/InlineNullValues.h:29-35) */
  int32_t encoding; /* mi355q_encoding */
  int32_t logical_type; /* SQL type after decoding (ENC_FIXED); 0 = same as `type`
                           (ENC_DICT implies INT32, ENC_DATE_IN_DAYS implies INT64) */
} mi355q_col_desc;

/* simple_quals / quals entry: `col <op> literal` (RelAlgExecutionUnit.h:170).  The quals of a plan are a CONJUNCTION of
 * such comparisons and of DISJUNCTIONS of them: quals whose `op` carries the same non-zero group number
 * (MI355Q_QUAL_IN_OR_GROUP(op, g), g = 1..3) are OR-ed together, the groups and the plain quals are AND-ed — `x < 5 AND
 * (y = 1 OR y = 2 OR z IS NULL)`.  With the reference's three-valued logic (logical_and / logical_or over nullable booleans,
 * LogicalIR.cpp:299-340) a row passes the filter when the whole condition is TRUE — `toBool`, :344-352: NULL counts as
 * false — i.e. when every plain qual is TRUE and every group has a member that is TRUE; NOT over a comparison of integers is
 * folded into the operator by the binding (NULL stays "not TRUE" either way; over DOUBLE / FLOAT operands a NaN would tell the
 * two apart).  Plans with a disjunction run in the row kernel.  Every other BOOLEAN conjunct — AND inside OR, NOT over a
 * disjunction, a BOOLEAN column, a long IN list — is a projected BOOLEAN expression (mi355q_expr: the NOT / AND / OR / IS NULL
 * micro-ops over comparisons) and the qual `that column = 1`, which keeps the step in the fast families; a conjunct that
 * holds an unsafe division — a DEFERRED qual in the reference, evaluated only for rows the other quals let through
 * (prioritizeQuals, LogicalIR.cpp:158-195) — is the second operand of a short-circuit AND inside such an expression.
 * A member of a group is never a `constrained_not_null` witness (OutputBufferInitialization.cpp:301-324 looks at top-level
 * conjuncts only). */
#define MI355Q_QUAL_OP(op) ((op) & 0xff)
#define MI355Q_QUAL_OR_GROUP(op) (((op) >> 8) & 0xff)
#define MI355Q_QUAL_IN_OR_GROUP(op, g) ((op) | ((g) << 8))
#define MI355Q_MAX_OR_GROUPS 3
typedef struct mi355q_qual {
  int32_t col; /* outer-table column index */
  int32_t op;  /* mi355q_op, + the disjunction it belongs to (MI355Q_QUAL_IN_OR_GROUP) */
  int64_t ival; /* literal for integer columns */
  double fval;  /* literal for double columns */
} mi355q_qual;

/* target_exprs entry (RelAlgExecutionUnit.h:173; Shared/TargetInfo.h:48-56) */
typedef struct mi355q_target {
  int32_t agg;   /* mi355q_agg */
  int32_t col;   /* argument column, -1 for COUNT(*); for PROJECT_KEY the index into
                    group_cols of the projected key (< 0 = the first) */
  int32_t table; /* 0 = outer (fact) column; 1 = inner (dim) column reached through
                    the join's row id */
  int32_t reserved;
  mi355q_qual cond; /* COUNT_IF / SUM_IF: the condition `outer col <op> literal` */
} mi355q_target;

/* ExpressionRange of a column (what getExpressionRange returns from chunk
 * metadata; ColRangeInfo GroupByAndAggregate.h / ExpressionRange.h). */
typedef struct mi355q_range {
  int32_t valid;     /* 0 = ExpressionRangeType::Invalid */
  int32_t has_nulls;
  int64_t min;
  int64_t max;
  double fp_min; /* used for double columns (keyless decisions only) */
  double fp_max;
  int64_t bucket; /* ExpressionRange::getBucket(): 0, or the stride of the values (86400 for
                     DATE columns, ExpressionRange.cpp:622-624); bucketed keys index the perfect
                     hash by (key - min) / bucket and never use the baseline or keyless
                     layouts (GroupByAndAggregate.cpp:344-349, QueryMemoryDescriptor.cpp:322-327) */
} mi355q_range;

typedef struct mi355q_join_table mi355q_join_table; /* opaque */

/* ---- projected expressions ("scan/filter/PROJECT"): a fixed parametric micro-op set, no JIT ----
 * What the reference compiles per query with CodeGenerator::codegen(expr) — casts (CastIR.cpp:21-57
 * codegenCast, :424-495 codegenCastBetweenIntTypes, :555-594 codegenCastToFp, :596-653 codegenCastFromFp),
 * + - * (ArithmeticIR.cpp:39 codegenArith, :187-262 codegenAdd, :264-340 codegenSub, :359-429 codegenMul,
 * :861-909 codegenBinOpWithOverflowForCPU) over columns and literals — is described here as a short POSTFIX
 * program per expression.  An expression is a VIRTUAL outer column: expression k is column index
 * n_cols + k wherever the plan takes an outer column (group_cols, targets[].col with table 0, quals[].col,
 * targets[].cond.col); n_cols + n_exprs <= MI355Q_MAX_COLS.  Expressions cannot be join keys.
 * Values carry SQL NULL in band, as the reference's do: the inline sentinel of the node's type
 * (Shared/InlineNullValues.h).  Integer + - * and narrowing casts are overflow-checked at the width of
 * the node's type; a row that passes every qual (and finds a match under an INNER join) and overflows
 * ends the step with MI355Q_ERR_OVERFLOW_OR_UNDERFLOW (ErrorCode 7), as the reference's row function
 * does; an expression inside a qual is evaluated for every row. */
typedef enum mi355q_expr_op {
  MI355Q_EX_COL = 1,   /* push outer column `arg` (decoded): type = the column's logical type,
                          nullable = the column's.  `arg` < n_cols: a physical column; n_cols + j with j < k inside
                          expression k: the VALUE of the plan's earlier expression j (a program longer than
                          MI355Q_MAX_EXPR_NODES is stated as several).  Like a column, that value exists for the row
                          whatever the position of the node: expression j is evaluated whenever k is, before it, and a
                          check that fires in j counts like one in k — so a subtree moves into an earlier expression
                          only from a place where it is evaluated unconditionally (not from a CASE branch or from the
                          second operand of a short-circuit AND / OR) */
  MI355Q_EX_LIT = 2,   /* push a literal of `type`: ilit (integers) / flit (DOUBLE, FLOAT); `reserved` = 1: the NULL of
                          `type` instead (a nullable value) */
  MI355Q_EX_CAST = 3,  /* cast the top of the stack to `type`.  integer -> wider integer: NULL to NULL
                          (cast_<from>_to_<to>_nullable, RuntimeFunctions.cpp:262-300); -> narrower
                          integer: error 7 when v > max(to) or v <= min(to)
                          (codegenCastBetweenIntTypesOverflowChecks, CastIR.cpp:497-553; NULL passes);
                          integer -> DOUBLE / FLOAT: sitofp; DOUBLE <-> FLOAT: fpext / fptrunc;
                          DOUBLE / FLOAT -> integer: round half away from zero, then truncate
                          (DEF_ROUND_NULLABLE, RuntimeFunctions.cpp:283-293); a value the integer type does not
                          hold is UNDEFINED in the reference (fptosi) and here: the conversion instruction's answer
                          (the host's and the device's differ), of which the expression's value keeps the type's
                          low bits */
  MI355Q_EX_ADD = 4,   /* pop rhs, pop lhs, push lhs + rhs; both operands have the node's `type`
                          (the analyzer has normalised them); NULL if either operand is NULL */
  MI355Q_EX_SUB = 5,
  MI355Q_EX_MUL = 6,
  MI355Q_EX_DIV = 7,   /* lhs / rhs (integers truncate towards zero; DOUBLE / FLOAT divide).  codegenDiv, ArithmeticIR.cpp:431-560
                          with g_null_div_by_zero off: when an operand may be NULL and one IS the type's NULL pattern the
                          zero check is skipped (codegenSkipOverflowCheckForNull, :343-357) and the result is
                          div_<type>_nullable[_lhs|_rhs] (RuntimeFunctions.cpp:46-71); otherwise a divisor equal to
                          zero (floating point: not "ordered and != 0", so NaN too) ends the step with
                          MI355Q_ERR_DIV_BY_ZERO (ErrorCode 1) for a row that counts, like an overflow */
  MI355Q_EX_MOD = 8,   /* lhs % rhs, integers only (codegenMod, :731-760): the divisor is tested against zero FIRST, whatever
                          the operands' NULLs (error 1); then mod_<type>_nullable[_lhs|_rhs] */
  /* COMPARISONS of two values (column vs column, column vs expression ...): pop rhs, pop lhs — both of ONE type, integer or
   * floating point, as the analyzer has normalised them — push a BOOLEAN, stored as MI355Q_INT8: 1 / 0, or NULL (the INT8
   * sentinel) when a NULLABLE operand is NULL (codegenCmp, CompareIR.cpp:230-330: plain icmp / fcmp for NOT NULL operands, else
   * <op>_<type>_nullable[_lhs|_rhs], RuntimeFunctions.cpp:73-107).  The node's `type` must be MI355Q_INT8.  A filter on a
   * comparison is a qual `expression column = 1` (TRUE; NULL is not TRUE, toBool LogicalIR.cpp:344-352) — or `IS NULL`. */
  MI355Q_EX_EQ = 9,
  MI355Q_EX_NE = 10,
  MI355Q_EX_LT = 11,
  MI355Q_EX_LE = 12,
  MI355Q_EX_GT = 13,
  MI355Q_EX_GE = 14,
  /* CASE WHEN cond THEN a ELSE b END (CodeGenerator::codegenCase, CaseIR.cpp:67-140): pop cond (a BOOLEAN as above), pop the
   * THEN value, pop the ELSE value — pushed in the order ELSE, THEN, cond, so that a chain of WHENs nests in the ELSE position
   * without growing the stack — push THEN if cond is TRUE, else ELSE (a NULL condition is not TRUE).  Both values have the
   * node's `type`; the result is nullable if either is.  The branches are LAZY, as the reference's basic blocks are: an
   * overflow or a division by zero in the branch that is not taken does not end the step (`CASE WHEN b <> 0 THEN a / b ELSE 0
   * END` never raises error 1).  A CASE without ELSE has the NULL literal there (MI355Q_EX_LIT with `reserved` = 1). */
  MI355Q_EX_CASE = 15,
  /* LOGIC over BOOLEAN values (MI355Q_INT8: 1 / 0 / the INT8 NULL); the node's `type` must be MI355Q_INT8.
   * NOT (CodeGenerator::codegenLogical(UOper), LogicalIR.cpp:363-379): pop a BOOLEAN; a NOT NULL operand gives !(v > 0)
   * (toBool :344-352), a nullable one logical_not (RuntimeFunctions.cpp:331-334): NULL stays NULL. */
  MI355Q_EX_NOT = 16,
  /* AND / OR (codegenLogical(BinOper), LogicalIR.cpp:299-342): pop rhs, pop lhs.  `reserved` = 0, the plain form: BOTH operands
   * are evaluated (a check that fires in either ends the step); NOT NULL operands give toBool(lhs) op toBool(rhs), otherwise the
   * three-valued logical_and / logical_or (RuntimeFunctions.cpp:336-358: NULL AND FALSE = FALSE, NULL OR TRUE = TRUE, else
   * NULL where an operand is NULL).  `reserved` = 1, the SHORT-CIRCUIT form the reference emits when an operand contains a
   * division whose divisor is not a non-zero constant (codegenLogicalShortCircuit :197-297 after contains_unsafe_division
   * :26-53; it swaps the operands so that the unsafe one is evaluated SECOND — the caller pushes them in that order): the
   * first operand alone decides where it can — NULL gives NULL (so here NULL AND FALSE = NULL, as in the reference's phi),
   * FALSE AND .. = FALSE, TRUE OR .. = TRUE — and the second is then NOT evaluated: its checks cannot fire
   * (`b <> 0 AND a / b > 1` never raises error 1); otherwise the result is the second operand's value (NULL if it is). */
  MI355Q_EX_AND = 17,
  MI355Q_EX_OR = 18,
  /* x IS NULL as a value (codegenIsNull, LogicalIR.cpp:381-432): pop a value of any type, push a NOT NULL BOOLEAN — 1 where a
   * NULLABLE operand equals its type's inline NULL (doubles / floats: ordered-equal to NULL_DOUBLE / NULL_FLOAT, the NULL
   * literal included), else 0.  An operand whose type is NOT NULL is not evaluated at all in the reference (constant
   * false): a check inside it cannot fire.  `x IS NOT NULL` is NOT(IS NULL(x)) (RelAlgTranslator.cpp:640-643). */
  MI355Q_EX_IS_NULL = 19,
  /* -x (codegenUMinus, ArithmeticIR.cpp:787-838): the node's `type` = the operand's.  Integers: a nullable operand equal to
   * the type's NULL stays NULL (uminus_<type>_nullable, RuntimeFunctions.cpp:247-258); otherwise the type's minimum — which
   * only a NOT NULL operand can hold as a value — ends the step with error 7.  DOUBLE / FLOAT: fneg, NULL stays NULL. */
  MI355Q_EX_UMINUS = 20
} mi355q_expr_op;

typedef struct mi355q_expr_node {
  int32_t op;   /* mi355q_expr_op */
  int32_t type; /* mi355q_type of the node's result (ignored for MI355Q_EX_COL) */
  int32_t arg;  /* MI355Q_EX_COL: outer column index: a physical column (< n_cols) or an earlier expression (n_cols + j) */
  int32_t reserved; /* MI355Q_EX_LIT: 1 = the NULL literal; MI355Q_EX_AND / _OR: 1 = the short-circuit form; else 0 */
  int64_t ilit;
  double flit;
} mi355q_expr_node;

typedef struct mi355q_expr {
  int32_t n_nodes; /* 1..MI355Q_MAX_EXPR_NODES, postfix; the stack never exceeds MI355Q_MAX_EXPR_STACK values and ends at 1 */
  int32_t reserved;
  mi355q_expr_node nodes[MI355Q_MAX_EXPR_NODES];
  mi355q_range range; /* getExpressionRange of the whole expression (ExpressionRange.cpp: casts keep the
                         operand's range, + - * combine the operands' bounds): drives the perfect-hash /
                         keyless decisions exactly like a column's range */
} mi355q_expr;

/* The subset of RelAlgExecutionUnit (RelAlgExecutionUnit.h:167-218) + the
 * expression ranges the planner derives from fragment metadata. */
typedef struct mi355q_plan {
  int32_t abi_version; /* MI355Q_ABI_VERSION */
  int32_t n_cols;      /* outer-table input columns (input_col_descs) */
  mi355q_col_desc cols[MI355Q_MAX_COLS];
  mi355q_range col_ranges[MI355Q_MAX_COLS]; /* of EVERY input column, not only the group keys: the keyless decisions read the
                                               aggregate arguments' ranges (get_keyless_info), and the index-partitioned family
                                               packs narrow records from them — as a HINT: a value outside its declared range
                                               costs time (a spill-list record), never correctness */
  int32_t n_inner_cols; /* inner (dim) table columns reachable via the join */
  mi355q_col_desc inner_cols[MI355Q_MAX_COLS];
  mi355q_range inner_col_ranges[MI355Q_MAX_COLS];

  int32_t n_quals; /* conjunction */
  mi355q_qual quals[MI355Q_MAX_QUALS];

  int32_t n_group_cols; /* 0 = non-grouped aggregate; 1..MI355Q_MAX_GROUP_COLS group-by
                           columns: integers of any mix of widths, or DOUBLE / FLOAT
                           (floating-point keys always take the baseline layout; the key is the
                           bit pattern of the value widened to double) */
  int32_t group_cols[MI355Q_MAX_GROUP_COLS];

  int32_t n_targets;
  mi355q_target targets[MI355Q_MAX_TARGETS];

  /* join_quals: one equi-join level  outer.col = inner.key [AND outer.col2 = inner.key2 ...] */
  int32_t join_outer_col;              /* first (or only) outer key column; -1 = no join */
  const mi355q_join_table* join_table; /* built by mi355q_join_build */
  int32_t n_join_cols;                 /* 0 or 1 = single-column key; 2..MI355Q_MAX_GROUP_COLS =
                                          composite key: join_outer_cols[i] pairs with the i-th
                                          inner key column the table was built from */
  int32_t join_outer_cols[MI355Q_MAX_GROUP_COLS];
  int32_t join_kind;                   /* mi355q_join_kind (JoinType, Shared/sqldefs.h) */
  int32_t reserved2;

  /* ExecutionOptions / globals that shape the layout */
  int64_t max_groups_buffer_entry_guess; /* baseline entry_count (Execute.cpp:111
                                            g_default_max_groups_buffer_entry_guess
                                            = 16384, or 2 x NDV estimate,
                                            RelAlgExecutor.cpp:4213-4218) */
  int32_t bigint_count;                  /* g_bigint_count: COUNT is BIGINT and slots are always
                                            8 bytes wide */
  int32_t output_columnar_hint;          /* mi355q_columnar_hint: 1 = the step's output buffer is
                                            COLUMNAR (output_columnar_hint of
                                            QueryMemoryDescriptor::init, g_enable_columnar_output;
                                            QueryMemoryDescriptor.cpp:311,387,515-531): one
                                            8-byte column per group column, then one column per
                                            slot, each align_to_int64(width * entry_count) bytes
                                            (getColOffInBytes :906-947).  Baseline key components
                                            are then 8 bytes wide (:387). */
  int64_t num_tuples;                    /* rows of the input tables (query_infos getNumTuples);
                                            0 = unknown / small.  A single-column GROUP BY whose
                                            targets are only COUNT(*) and projections of a key of
                                            at most 4 bytes gets 4-BYTE slots while this stays
                                            <= UINT32_MAX and bigint_count is off
                                            (pick_target_compact_width,
                                            QueryMemoryDescriptor.cpp:748-840) */
  /* projected expressions: virtual outer columns n_cols .. n_cols + n_exprs - 1 (see mi355q_expr) */
  int32_t n_exprs;
  int32_t reserved3;
  mi355q_expr exprs[MI355Q_MAX_EXPRS];
  /* Projection steps: RelAlgExecutionUnit::scan_limit (RelAlgExecutionUnit.h:178) — LIMIT + OFFSET of a projection
   * without ORDER BY; 0 = none.  With a limit the output buffer has exactly that many entries and the step ends
   * normally when more rows match (the first scan_limit of them, in (fragment, row) order, are kept; the reference stops
   * its loop at max_matched, QueryTemplateGenerator.cpp:751-780).  Without one, a row that finds the buffer full ends
   * the step with a NEGATIVE code — get_scan_output_slot returns NULL and the row function answers -pos
   * (GroupByAndAggregate.cpp:1151-1156) — here -(number of matching rows), clamped to -(INT32_MAX - 64): the caller re-runs with
   * max_groups_buffer_entry_guess = that count (the reference sizes the buffer by a COUNT(*) pre-flight,
   * RelAlgExecutor::getFilteredCountAll, or doubles the guess). */
  int64_t scan_limit;
} mi355q_plan;

/* plan.output_columnar_hint */
typedef enum mi355q_columnar_hint {
  MI355Q_OUTPUT_ROWWISE = 0,
  MI355Q_OUTPUT_COLUMNAR = 1,
  /* the layout DECISIONS of a columnar descriptor (8-byte baseline key components) stored
   * row-wise: the library's own intermediate form of a columnar step, also accepted from callers */
  MI355Q_OUTPUT_ROWWISE_COLUMNAR_DECISIONS = 2
} mi355q_columnar_hint;

/* QueryMemoryDescriptor mirror (Descriptors/QueryMemoryDescriptor.h).  Row-wise layout unless
 * output_columnar is set; slots
 * are 8 bytes wide (crt_min_byte_width = 8, Execute.cpp:2237) except for the COUNT(*)-only
 * shapes that pick_target_compact_width narrows to 4 (slot_width). */
typedef struct mi355q_qmd {
  int32_t desc_type; /* mi355q_desc_type */
  int32_t keyless;   /* keyless_hash_ */
  int32_t idx_target_as_key; /* slot index whose value != init marks a live entry */
  int32_t key_width; /* getEffectiveKeyWidth(): 8, or 4 when every baseline key component
                        fits int32 (pick_baseline_key_width, QueryMemoryDescriptor.cpp:135-146) */
  int32_t group_col_count;
  int32_t slot_count;
  int64_t entry_count;
  int64_t min_val; /* single-column perfect hash: col_range_info.min */
  int64_t max_val; /* single-column perfect hash: col_range_info.max (NULL key maps to max+1);
                      multi-column perfect hash: the cardinality product (= entry_count), as
                      getColRangeInfo returns it (GroupByAndAggregate.cpp:268-273) */
  int64_t bucket;
  /* per group column (perfect hash): range minimum, getBucketedCardinality, and the value a
   * NULL key is translated to (max + max(bucket, 1); GroupByAndAggregate.cpp:1339-1345).  The entry
   * index of a multi-column key is sum_i (key_i - min_i) / bucket_i * prod_{j<i} card_j
   * (codegenPerfectHashFunction, GroupByAndAggregate.cpp:1546-1598). */
  int64_t group_min[MI355Q_MAX_GROUP_COLS];
  int64_t group_card[MI355Q_MAX_GROUP_COLS];
  int64_t group_null_key[MI355Q_MAX_GROUP_COLS];
  int64_t group_bucket[MI355Q_MAX_GROUP_COLS];
  int32_t group_has_nulls[MI355Q_MAX_GROUP_COLS];
  int32_t has_nulls;
  int32_t row_size;      /* bytes, getRowSize() (QueryMemoryDescriptor.cpp:848):
                            align8(key bytes) + align8(slot_count * slot_width).  For a
                            columnar descriptor: the size of one entry in the row-wise form of
                            the same decisions (the buffer itself is sized by
                            mi355q_qmd_buffer_bytes) */
  int32_t slot_width;    /* 8, or 4 (ColSlotContext::setAllSlotsPaddedSize(min_slot_size)) */
  int32_t output_columnar; /* output_columnar_: the buffer is
                              [group col 0 | ... | slot 0 | slot 1 | ...], every column
                              align_to_int64(width * entry_count) bytes, group columns 8 bytes
                              wide (getPrependedGroupColOffInBytes :962-975), none when
                              keyless; see mi355q_qmd_group_col_offset / _slot_col_offset */
  int32_t key_bytes;     /* align_to_int64(group_col_count * key_width), 0 if keyless */
  int32_t n_targets;
  int32_t target_slot[MI355Q_MAX_TARGETS];      /* first slot of each target; -1 if the
                                                   target is read from the key columns
                                                   (target_groupby_indices) */
  int32_t target_key_idx[MI355Q_MAX_TARGETS];   /* PROJECT_KEY: which group column */
  int32_t target_skip_null[MI355Q_MAX_TARGETS]; /* TargetInfo.skip_null_val */
  int32_t target_is_fp[MI355Q_MAX_TARGETS];     /* result is double */
  int32_t target_agg[MI355Q_MAX_TARGETS];       /* mi355q_agg */
  int32_t target_arg_is_fp[MI355Q_MAX_TARGETS]; /* slot holds double bits (SUM/MIN/MAX/AVG
                                                   of a double column) */
  int32_t target_arg_is_f32[MI355Q_MAX_TARGETS]; /* slot holds FLOAT bits in its low 4 bytes
                                                    (SUM/MIN/MAX/AVG of a float column) */
  int64_t target_null[MI355Q_MAX_TARGETS];      /* bit pattern of the result type's NULL
                                                   (null_val_bit_pattern,
                                                   ResultSetBufferAccessors.h:229); EMPTY_KEY_64
                                                   (never a key) for the projection of a NOT
                                                   NULL key column, which is never NULL
                                                   (ResultSet::isNull tests the type first) */
  int64_t init_vals[MI355Q_MAX_SLOTS];          /* init_agg_val_vec
                                                   (OutputBufferInitialization.cpp:24) */
  int32_t slot_bytes[MI355Q_MAX_SLOTS];         /* getPaddedSlotWidthBytes(slot): slot_width for every slot, except in a
                                                   COLUMNAR PROJECTION, whose slot columns have the targets' logical
                                                   widths (1 / 2 / 4 / 8) */
} mi355q_qmd;

/* FetchResult mirror (ColumnFetcher.h:46-49): borrowed device pointers. */
typedef struct mi355q_inputs {
  int32_t device_id;
  int32_t n_frags;
  /* col_buffers[frag * n_cols + col] -> device pointer to a dense fixed-width chunk */
  const void* const* col_buffers;
  const int64_t* num_rows; /* per fragment (host array) */
  /* inner table, linearized into one chunk per column (reference fetches inner
   * tables as one "all fragments" buffer, ColumnFetcher::getAllTableColumnFragments) */
  const void* const* inner_col_buffers; /* [n_inner_cols] */
  int64_t inner_num_rows;
  /* Generation of the inner columns' CONTENT.  A join table keeps, per inner column, payloads derived from the
   * column's values (per-key counts / sums for the payload probes of large outer tables).  They are reused
   * while (device pointer, inner_version) are unchanged: a caller that rewrites an inner column in place — or
   * frees it and lets a new column land on the same address — must pass a different inner_version (or call
   * mi355q_join_invalidate_payload); 0 is a valid version like any other. */
  int64_t inner_version;
} mi355q_inputs;

typedef struct mi355q_result mi355q_result; /* opaque: QueryMemoryDescriptor + buff_ */

/* ExecutionOptions-like knobs for this library. */
typedef struct mi355q_exec_options {
  void* stream;      /* hipStream_t to launch on; NULL = library-owned stream */
  void* out_buffer;  /* optional caller-owned device buffer for the result storage
                        (>= mi355q_qmd_buffer_bytes); NULL = library allocates */
  int32_t force_generic; /* 1 = always use the generic row kernel (testing) */
  int32_t kernel_variant; /* 0 = plan-time choice; >0 selects a specific variant of
                             the chosen family (testing / tuning) */
  int64_t scratch_bytes;  /* cap for partition scratch (0 = default) */
  /* testing / tuning knobs: explicit fields of the ABI (nothing in the library reads the environment);
   * 0 = the built-in choice */
  int32_t tune_blocks_per_cu;   /* workgroups per CU of the streaming kernels (experiments) */
  int32_t probe_keyed_passes;   /* keyed payload probe: passes per partition, 1..4 (tests: several passes
                                   on a small table) */
  int64_t pass_rows;            /* packed-key and projected-expression routes: rows per pass (tests force
                                   several passes; the fragments are never split) */
  uint32_t flags;               /* MI355Q_OPT_* */
  int32_t tune_cus;             /* plan and launch as if the device had this many CUs (experiments: how a family
                                   scales with the CU count, whether two families could share the device) */
  int32_t tune_overlap_cus;     /* partitioned GROUP BY: > 0 = phase 1 (k_part_scatter) of chunk i + 1 runs on this many
                                   CUs WHILE phase 2 (k_part_aggregate) of chunk i runs on the others (second stream,
                                   two record buffers of half the scratch each); 0 = the built-in choice, -1 = phases
                                   one after the other on the whole device */
  int32_t reserved0;            /* 0 */
} mi355q_exec_options;
#define MI355Q_OPT_TRACE 1u              /* host-side wall-clock marks and phase-2 cycle counters on stderr */
#define MI355Q_OPT_NO_PAIR_RENDEZVOUS 2u /* partitioned GROUP BY phase 2: no rendezvous of the sub-range pair */
#define MI355Q_OPT_PROBE_NO_PACING 4u    /* L2 payload probe: no per-XCD partition pacing */
#define MI355Q_OPT_NO_LDS_BASELINE 8u    /* baseline layouts: do not try the few-groups LDS member (the library sets
                                            this itself when it re-runs a step whose groups did not fit a replica) */
#define MI355Q_OPT_LDS_BASELINE_LARGE 16u /* ... try it with the largest replica LDS holds (second attempt: the first
                                            uses 256-slot replicas, many of them, for tables with a handful of groups) */
#define MI355Q_OPT_LDS_BASELINE_WINDOWS 32u /* ... third attempt: the groups spread over 8 windows (classes of a key hash),
                                            one workgroup per window and row stripe, the largest replica each */
#define MI355Q_OPT_LDS_GENERIC_MEMBER 64u /* few-groups LDS GROUP BY: the run-time-role member even where a typed member
                                            (roles compiled in) applies (tests compare the two) */
#define MI355Q_OPT_NO_IDX_PART 128u       /* large perfect-hash tables: not the index-partitioned family (the library sets this
                                            itself when it re-runs a step whose spill list overflowed) */
#define MI355Q_OPT_NO_COMPILED_FILTER 256u /* BOOLEAN filters: the interpreter pass (k_project) even where the filter compiles
                                            into atoms + a truth table (tests and tools/bool_filter_bench.py compare the two) */
#define MI355Q_OPT_FILTER_PREPASS 512u    /* compiled filters with program atoms (arithmetic / two-column leaves): the row-mask
                                            pre-pass (k_filter_mask) even where the typed few-groups member evaluates the
                                            atoms itself; with MI355Q_OPT_LDS_GENERIC_MEMBER the pre-pass's general member
                                            (tests and tools/bool_filter_bench.py compare them) */
#define MI355Q_OPT_NO_IDX_PACK 1024u      /* index-partitioned family: the plain 4 / 8 / 16-byte records even where the value
                                            columns' ranges allow the packed 2- or 4-byte word (tests and tools/refbench.py
                                            compare the two) */

/* per-call timing/selection report (what launchGpuCode logs,
 * QueryExecutionContext.cpp:334,364,579) */
typedef struct mi355q_exec_report {
  char kernel_name[64]; /* dominant kernel chosen at plan time */
  float kernel_ms;      /* HIP-event time summed over the launches of the dominant kernel,
                           measured on the launch stream; avg = kernel_ms / n_launches */
  float total_ms;       /* HIP-event time of the whole call on the launch stream */
  int32_t n_launches;   /* launches of the dominant kernel */
  int32_t variant;      /* member of the family that ran (informational; e.g. k_groupby_lds: 4 run-time roles, 5 typed;
                           k_idx_scatter: 6 plain records, 7 / 8 the packed 4- / 2-byte word) */
  int64_t rows_scanned;
  int64_t algorithmic_bytes; /* column bytes the plan must read */
  int64_t spilled_rows;      /* rows that took the direct-atomic spill path */
} mi355q_exec_report;

/* ---- library ---- */
int32_t mi355q_abi_version(void);
/* sizeof() of the ABI structs as compiled into the library, for binding self-checks:
 * 1 plan, 2 qmd, 3 inputs, 4 exec_options, 5 exec_report, 6 join_spec */
int64_t mi355q_abi_sizeof(int32_t which);
const char* mi355q_error_string(int32_t code);
int32_t mi355q_device_count(void);
/* The library keeps a per-device workspace between calls (partition scratch, fragment
 * tables, timing events) the way the reference keeps its per-device allocator arenas
 * (CudaAllocator, DataMgr/Allocators/CudaAllocator.cpp); this frees it. */
int32_t mi355q_release_workspace(int32_t device_id);
/* fills name (<=256 bytes) and basic properties of a device */
int32_t mi355q_device_info(int32_t device_id, char* name, int32_t* cu_count,
                           int64_t* total_mem, int64_t* free_mem, int32_t* mem_clock_khz,
                           int32_t* mem_bus_width);

/* ---- plan -> layout ---- */
int32_t mi355q_qmd_init(const mi355q_plan* plan, mi355q_qmd* out);
int64_t mi355q_qmd_buffer_bytes(const mi355q_qmd* qmd);
/* Columnar descriptors: byte offset of group column g (getPrependedGroupColOffInBytes,
 * QueryMemoryDescriptor.cpp:962-975) and of slot column s (getColOffInBytes :906-929) in the
 * buffer; -1 if the descriptor is row-wise or the index is out of range (a keyless descriptor
 * has no group columns). */
int64_t mi355q_qmd_group_col_offset(const mi355q_qmd* qmd, int32_t g);
int64_t mi355q_qmd_slot_col_offset(const mi355q_qmd* qmd, int32_t s);

/* ---- execute one query step on one device ---- */
int32_t mi355q_execute(const mi355q_plan* plan, const mi355q_inputs* inputs,
                       const mi355q_exec_options* opts, mi355q_result** out,
                       mi355q_exec_report* report);

/* Stream-ordered execute (the reference's launch is synchronous, DeviceKernel.cpp:84; one process per GPU with
 * collectives behind the step wants the opposite): every kernel of the step is enqueued on opts->stream (or the
 * library's stream) and the call returns without waiting for the device.  *out may be handed at once to
 * stream-ordered consumers ON THE SAME STREAM (mi355q_shard_pads, mi355q_shard_merge_*, a collective enqueued
 * behind them); everything that needs the host — the error code, the re-run with the direct member after a
 * spill overflow, the report — happens in mi355q_wait, which must be called exactly once per pending handle
 * (it frees it).  On an error from mi355q_wait the result handle is still the caller's to free.  One step per
 * device can be in flight; any other mi355q_execute[_async] on the device finishes it first.  Routes that need
 * the host mid-step (projected expressions, packed multi-column keys, columnar / 4-byte-slot results, the
 * first build of a join payload) complete inside the call. */
typedef struct mi355q_pending mi355q_pending;
int32_t mi355q_execute_async(const mi355q_plan* plan, const mi355q_inputs* inputs,
                             const mi355q_exec_options* opts, mi355q_result** out,
                             mi355q_pending** pending);
int32_t mi355q_wait(mi355q_pending* pending, mi355q_exec_report* report);

/* Allocates (and keeps, until mi355q_release_workspace) what a step of this plan over inputs of this shape needs
 * from the per-device workspace — the partition scratch above all: tens of GB for a 10 B-row GROUP BY, a
 * hipMalloc of that size takes 1 - 2 s — so that the first mi355q_execute does not pay for it.  `inputs` only
 * has to carry device_id, n_frags, num_rows (and the join's inner_num_rows); the column pointers may be NULL. */
int32_t mi355q_reserve_workspace(const mi355q_plan* plan, const mi355q_inputs* inputs,
                                 const mi355q_exec_options* opts, int64_t* reserved_bytes);

/* EXPLAIN for one step (what the reference prints for `EXPLAIN <query>`: the generated kernel; here: which members of
 * the fixed family run — QueryEngine/RelAlgExecutor.cpp:executeRelAlgQuery just_explain, Execute.cpp:2107
 * ExecutionOptions::just_explain).  The route mi355q_execute would take for this plan over inputs of this shape, as a
 * " > "-separated chain of derived-plan stages and the kernel family that runs the step, e.g.
 *   "k_project > k_pack_keys (entry index, baseline temp) > k_part_scatter + k_part_aggregate > k_unpack_emit".
 * Nothing is launched and nothing is allocated; `inputs` as for
 * mi355q_reserve_workspace.  *scratch_bytes = the partition scratch the step would ask for.  The answer depends on the
 * input size (small inputs take the row kernel, large ones the partitioned / packed members) and on the device's CUs. */
int32_t mi355q_explain(const mi355q_plan* plan, const mi355q_inputs* inputs, const mi355q_exec_options* opts,
                       char* route, int64_t route_len, int64_t* scratch_bytes);

/* ---- result set ---- */
int32_t mi355q_result_create(const mi355q_qmd* qmd, int32_t device_id, void* device_buffer,
                             mi355q_result** out); /* wraps/allocates + initialises */
/* wraps an existing, already populated device buffer of this layout (no initialisation;
 * the caller keeps ownership) — e.g. a partial buffer received from another GPU */
int32_t mi355q_result_wrap(const mi355q_qmd* qmd, int32_t device_id, void* device_buffer,
                           mi355q_result** out);
void mi355q_result_free(mi355q_result* r);
int32_t mi355q_result_qmd(const mi355q_result* r, mi355q_qmd* out);
void* mi355q_result_device_ptr(const mi355q_result* r);
int64_t mi355q_result_bytes(const mi355q_result* r);
int32_t mi355q_result_copy_to_host(const mi355q_result* r, void* dst, int64_t dst_bytes);
/* this += that, slot-wise with the targets' aggregate functions; baseline layouts
 * re-hash every live entry of `that` into `this` (ResultSetReduction.cpp:203-383,
 * :783-826).  Both on the same device. */
int32_t mi355q_result_reduce(mi355q_result* this_rs, const mi355q_result* that_rs,
                             void* stream);
/* Projection results: this = this's rows followed by that's (ResultSet::append, ResultSet.cpp:307-335, as
 * Executor::resultsUnion puts the devices' results together, Execute.cpp:1642-1694): the entry count becomes the sum of
 * both, total_matched too; `this` moves into a buffer the library owns (mi355q_result_device_ptr changes — a caller-
 * provided out_buffer is left behind, not freed).  mi355q_result_reduce does the same when both sides are projections.
 * MI355Q_ERR_UNSUPPORTED for a wrapped buffer (its rows need not sit at the front). */
int32_t mi355q_result_append(mi355q_result* this_rs, const mi355q_result* that_rs, void* stream);
/* number of non-empty entries (ResultSet::rowCount, ResultSet.h:327) */
int64_t mi355q_result_row_count(const mi355q_result* r);
/* Projection results: the rows that passed the quals (the kernel's total_matched word, KernelParam::TOTAL_MATCHED,
 * enums.h:62-77); > entry_count when a scan_limit cut the output.  -1 for any other result. */
int64_t mi355q_result_total_matched(const mi355q_result* r);
/* Materialise rows the way ResultSet::getNextRow does, in entry order: for every
 * non-empty entry one row of n_targets values.  Integer-typed targets land in ival,
 * double-typed in dval (AVG = sum/count, NULL if count == 0); is_null flags SQL NULL.
 * Arrays are [max_rows * n_targets].  Returns rows written in *n_rows. */
int32_t mi355q_result_fetch_rows(const mi355q_result* r, int64_t max_rows, int64_t* ival,
                                 double* dval, int8_t* is_null, int64_t* n_rows);

/* ORDER BY <target> [ASC | DESC] [NULLS FIRST | LAST] LIMIT k over a grouped (or projected)
 * result, on the device: the k best non-empty entries are written to out_rows_dev as whole rows
 * of r's layout, in order; *n_rows = min(k, live entries).  k <= 4096.  Replaces
 * ResultSet::sort -> baselineSort (ResultSet.cpp:801-851) -> baseline_sort
 * (ResultSetSortImpl.cu) / TopKSort.cu for one order entry; ties at the k-th position are
 * broken arbitrarily, as in the reference.  AVG targets are ordered by sum / count. */
int32_t mi355q_result_topk(const mi355q_result* r, int32_t target_idx, int32_t descending,
                           int32_t nulls_first, int64_t k, void* out_rows_dev, int64_t* n_rows,
                           void* stream);

/* ORDER BY <target> [ASC | DESC] [NULLS FIRST | LAST], ... LIMIT limit OFFSET offset over a grouped
 * result, on the device, any number of order entries and any limit (0 = every live row): a stable
 * least-significant-first radix sort of the entry permutation by each order entry's 64-bit
 * order-preserving key.  Replaces ResultSet::sort (ResultSet.cpp:781-851: baselineSort /
 * radixSortOnGpu / parallelTop; comparator ResultSetComparator::operator(), :1310-1470) and
 * ResultSetSortImpl.cu / TopKSort.cu / StreamingTopN for sorts the top-k selection above does not
 * take.  out_rows_dev receives whole rows of r's row-wise layout in order (min(limit, live - offset)
 * rows, the count in *n_rows; it must have room for `limit` rows, or for mi355q_result_row_count(r)
 * rows when limit is 0).  Rows that tie on every order entry come out in an unspecified order, as
 * in the reference. */
typedef struct mi355q_order_entry {
  int32_t target_idx;  /* tle_no - 1 */
  int32_t descending;  /* is_desc */
  int32_t nulls_first;
  int32_t reserved;
} mi355q_order_entry;
int32_t mi355q_result_sort(const mi355q_result* r, const mi355q_order_entry* order, int32_t n_order,
                           int64_t limit, int64_t offset, void* out_rows_dev, int64_t* n_rows, void* stream);

/* ColumnarResults for a grouped result, on the device (QueryEngine/ColumnarResults.cpp:1374-1600
 * materializeAllColumnsGroupBy: locateAndCountEntries -> partial sums -> compactAndCopyEntries):
 * the non-empty entries, in entry order, as one dense 8-byte column per target.  Integer targets
 * are written as int64, floating-point ones as double (AVG = sum / count; FLOAT results widened),
 * SQL NULL as the inline sentinel (mi355q_qmd.target_null for integers, NULL_DOUBLE for
 * floating point).  cols_dev[t] must hold mi355q_result_row_count(r) values; *n_rows receives the
 * row count.  The rows come out in the order mi355q_result_fetch_rows iterates them. */
int32_t mi355q_result_to_columns(const mi355q_result* r, void* const* cols_dev, int32_t n_cols,
                                 int64_t* n_rows, void* stream);

/* ---- join hash tables ---- */
typedef struct mi355q_join_spec {
  int32_t device_id;
  int32_t key_type;     /* mi355q_type of the (first) inner key column */
  int32_t key_nullable;
  int32_t prefer_baseline; /* 1 = skip the perfect attempt (testing keyed tables) */
  const void* key_buffer; /* device pointer, inner key column, linearized */
  int64_t num_rows;
  mi355q_range key_range; /* inner key ExpressionRange (single-column keys) */
  int64_t max_perfect_entries; /* 0 = default (PerfectJoinHashTable.cpp:219-224) */
  /* composite keys (inner_outer_pairs_.size() > 1 -> BaselineJoinHashTable): further inner
   * key columns after the first; n_keys 0 or 1 = single column */
  int32_t n_keys;
  int32_t one_to_many; /* 0 = OneToOne only: a duplicate key fails with
                          MI355Q_ERR_JOIN_NOT_ONE_TO_ONE; 1 = rebuild as OneToMany like
                          HashJoin::getInstance does when the OneToOne fill reports a
                          duplicate (PerfectJoinHashTable.cpp:255-290 reify ->
                          NeedsOneToManyHash); 2 = build OneToMany straight away */
  int32_t more_key_types[MI355Q_MAX_GROUP_COLS - 1];
  int32_t more_key_nullables[MI355Q_MAX_GROUP_COLS - 1];
  const void* more_key_buffers[MI355Q_MAX_GROUP_COLS - 1];
  int64_t keyed_entry_count; /* 0 = 2 x num_rows; the reference sizes keyed tables at 2 x its
                                (HyperLogLog) estimate of the distinct keys,
                                BaselineJoinHashTable.cpp:484-486 — a binding that has that
                                estimate passes 2 x NDV here */
} mi355q_join_spec;

/* HashType / layout of the buffer (docs/source/execution/hash_joins.rst "Hash Join Buffers"):
 *   0 OneToOne perfect   int32 slot[max-min+1], -1 empty (HashJoinRuntime.cpp:71-86)
 *   1 OneToOne keyed     entry_count x (key components..., payload) of 4- or 8-byte integers,
 *                        first component EMPTY (INT32_MAX / INT64_MAX) when free, payload = row
 *                        id (HashJoinRuntime.cpp:346-373, :505-538); component width 8 iff an
 *                        inner key column is wider than 4 bytes
 *                        (BaselineJoinHashTable::getKeyComponentWidth), entry_count = 2 x rows
 *   2 OneToMany perfect  int32 offsets[entries] | int32 counts[entries] | int32 payloads[rows]:
 *                        offsets -1 where no row has the key, else the start of the key's row
 *                        ids in payloads (fill_one_to_many_hash_table,
 *                        HashJoinRuntime.cpp:1503-1560: count_matches -> inclusive_scan ->
 *                        fill_row_ids)
 *   3 OneToMany keyed    keys[entries x components] | offsets | counts | payloads
 *                        (fill_one_to_many_baseline_hash_table, HashJoinRuntime.cpp:1975-2100)
 * The order of the row ids inside one key's payload run is build-order dependent in the
 * reference as well (hash_joins.rst "comparing buffers"). */
int32_t mi355q_join_build(const mi355q_join_spec* spec, void* stream,
                          mi355q_join_table** out);
void mi355q_join_free(mi355q_join_table* t);
int32_t mi355q_join_info(const mi355q_join_table* t, int32_t* hash_type, int64_t* entry_count,
                         int64_t* min_key, int64_t* max_key, void** device_ptr,
                         int64_t* bytes, float* build_ms);
/* Drops every payload the table caches about inner columns (see mi355q_inputs.inner_version); the next step
 * that wants one rebuilds it. */
int32_t mi355q_join_invalidate_payload(mi355q_join_table* t);
/* What the payload cache holds: device bytes, the time the last build took (a first-run cost of the payload
 * probes that is NOT inside any step's kernel_ms), and the inner_version it was built for. */
int32_t mi355q_join_payload_info(const mi355q_join_table* t, int64_t* bytes, float* build_ms, int64_t* inner_version);
/* key component count and width (bytes) of a keyed table (1 / 8 for perfect tables) */
int32_t mi355q_join_key_shape(const mi355q_join_table* t, int32_t* key_components,
                              int32_t* component_width);

/* ---- Arrow export (ArrowResultSetConverter::convertToArrow, QueryEngine/ArrowResultSetConverter.cpp):
 * the rows of a result as ONE struct-typed ArrowArray (a record batch) + its ArrowSchema through the
 * Arrow C Data Interface — plain C structs defined by the Arrow specification, no Arrow library on
 * either side of the ABI.  One child per target: BIGINT / COUNT / projected integer keys -> int64 ("l"),
 * DOUBLE / AVG / floating-point targets -> float64 ("g"), SQL NULL -> validity bitmap.  The columns are
 * produced dense on the device (mi355q_result_to_columns), copied to the host once, and owned by the
 * exported array until its release callback runs.  Non-grouped results export their single row.
 * The struct definitions below are the specification's (arrow/c/abi.h); they are guarded so a
 * translation unit that already includes Arrow's header can include this one too. */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif
/* names: n_targets column names, or NULL for "target_<i>". */
int32_t mi355q_result_export_arrow(const mi355q_result* r, const char* const* names, struct ArrowSchema* out_schema,
                                   struct ArrowArray* out_array, void* stream);

/* ---- multi-device merge helpers (one process per GPU; the collective itself is
 * issued by the host with RCCL between these calls) ---- */
/* Baseline layouts: compact the live entries of `r` into `n_parts` contiguous runs of
 * whole rows (row_size bytes each), run p holding the keys with
 * MurmurHash3(key) / entry-hash % n_parts == p.  out_rows must hold entry_count rows;
 * part_counts (device, int64[n_parts]) receives the run lengths. */
int32_t mi355q_shard_partition(const mi355q_result* r, int32_t n_parts, void* out_rows,
                               int64_t* part_counts_dev, void* stream);
/* Slice exchange (row-wise baseline tables with one 8-byte key and 8-byte slots): rank r of `world`
 * owns the keys whose home slot MurmurHash3(key) % entry_count (GroupByRuntime.cpp:20-48) lies in
 * [r * entry_count / world, (r + 1) * entry_count / world).  Linear probing keeps a key at or just
 * after its home slot, so what rank r needs from a peer is that peer's table rows of the same range —
 * sent in place, every split size known without a count exchange — plus the `pad_rows` rows after the
 * range's end (the tail of a probe cluster that crosses the boundary; wraps at the table's end).
 * mi355q_shard_pads copies the pad after each of the `world` ranges into out_pads_dev
 * ([world][pad_rows] whole rows) and sets ok_dev[r] = 1 when pad r contains an empty slot, i.e. no
 * cluster starting in range r reaches beyond its pad (otherwise fall back to mi355q_shard_partition).
 * mi355q_shard_merge_range folds `n_rows` received rows into r, keeping only the keys whose home slot
 * is in [home_lo, home_hi) (a slice also contains strays of the previous range's clusters, a pad
 * contains rows of the next range). */
int32_t mi355q_shard_pads(const mi355q_result* r, int32_t world, int32_t pad_rows, void* out_pads_dev,
                          int32_t* ok_dev, void* stream);
int32_t mi355q_shard_merge_range(mi355q_result* r, const void* rows, int64_t n_rows, int64_t home_lo,
                                 int64_t home_hi, void* stream);
/* The same fold in one launch and in LDS (phase 2 of the partitioned GROUP BY without records): rows
 * [home_lo, home_hi) of `n_src` tables of r's layout — slices[i] is the device address of row home_lo of
 * table i, i.e. what peer i sent — plus the pad_rows rows that followed each (pads[i]; pads may be NULL)
 * are merged into rows [home_lo, home_hi) of r, which are OVERWRITTEN (r is a fresh table in the slice
 * exchange).  n_src <= 16.  MI355Q_ERR_UNSUPPORTED: a slot program the partitioned family does not take
 * (r untouched: use mi355q_shard_merge_range); MI355Q_ERR_OUT_OF_SLOTS: the table or the stray list
 * overflowed (r incomplete: re-create it and use mi355q_shard_merge_range). */
int32_t mi355q_shard_merge_slices(mi355q_result* r, const void* const* slices, const void* const* pads,
                                  int32_t n_src, int32_t pad_rows, int64_t home_lo, int64_t home_hi,
                                  void* stream);
/* Insert `n_rows` whole rows (same layout as r) into r with the reduce semantics. */
int32_t mi355q_shard_merge_rows(mi355q_result* r, const void* rows, int64_t n_rows,
                                void* stream);

/* ---- synthetic column generators (same splitmix64 stream as
 * oracle/oracle.cpp) — device-side so 10 B-row tables never touch the host ---- */
typedef enum mi355q_gen_kind {
  MI355Q_GEN_I32_UNIFORM31 = 1, /* (int32)(u >> 33) in [0, 2^31) */
  MI355Q_GEN_I32_MOD = 2,       /* (int32)(u % a) + b */
  MI355Q_GEN_I64_MOD = 3,       /* (int64)(u % a) + b */
  MI355Q_GEN_I64_MOD_MUL = 4,   /* (int64)(u % a) * b + c */
  MI355Q_GEN_F64_UNIT = 5       /* (double)(u >> 11) * 2^-53 * a_f */
} mi355q_gen_kind;
int32_t mi355q_generate_column(int32_t device_id, void* dst, int64_t n_rows, int64_t row_offset,
                               int32_t kind, uint64_t seed, int64_t a, int64_t b, int64_t c,
                               double a_f, int32_t null_every, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355Q_H */
