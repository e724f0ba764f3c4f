// heavydb_mock.h — MINIMAL STAND-INS for the HeavyDB types integration/Mi355qExecutor.cpp touches.
//
// HeavyDB cannot be built in this environment (Thrift, LLVM, Boost, Arrow, TBB ... are absent), so
// the binding TU is compiled against these declarations instead of the reference's headers.  They restate the
// INTERFACE the binding uses — same namespaces, class names, accessor names and argument meaning — and nothing of
// the implementations; a maintainer drops this header and includes the real ones:
//   SQLTypeInfo, SQLTypes, EncodingType        Shared/sqltypes.h
//   SQLOps, SQLAgg, JoinType                   Shared/sqldefs.h
//   Datum                                      Shared/Datum.h
//   Analyzer::Expr / ColumnVar / Constant /    Analyzer/Analyzer.h
//     UOper / BinOper / AggExpr
//   InputDescriptor, InputColDescriptor        QueryEngine/Descriptors/InputDescriptors.h
//   RelAlgExecutionUnit, JoinCondition         QueryEngine/RelAlgExecutionUnit.h:155-218
//   FetchResult                                QueryEngine/ColumnFetcher.h:40-49
//   ExpressionRange, getExpressionRange        QueryEngine/ExpressionRange.h
//   QueryMemoryDescriptor (accessors only)     QueryEngine/Descriptors/QueryMemoryDescriptor.h
//   ResultSet, ResultSetStorage                QueryEngine/ResultSet.h, ResultSetStorage.h
//   QueryExecutionError                        QueryEngine/ErrorHandling.h
//   Executor (getCudaStream, getRowSetMemoryOwner)   QueryEngine/Execute.h
#pragma once

#include <cstdint>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

// ---- Shared/sqltypes.h:65-100, :261-273, Shared/sqldefs.h:31-58, :76-90: the enumerators this binding names, with the
// reference's numeric values (tests/test_integration_glue.py::test_mock_enums_match_the_reference compiles a TU that
// includes BOTH this file's values and the reference's headers and static_asserts every pair)
enum SQLTypes {
  kNULLT = 0, kBOOLEAN = 1, kCHAR = 2, kVARCHAR = 3, kNUMERIC = 4, kDECIMAL = 5, kINT = 6, kSMALLINT = 7, kFLOAT = 8,
  kDOUBLE = 9, kTIME = 10, kTIMESTAMP = 11, kBIGINT = 12, kTEXT = 13, kDATE = 14, kARRAY = 15, kPOINT = 18, kTINYINT = 22
};
enum EncodingType { kENCODING_NONE = 0, kENCODING_FIXED = 1, kENCODING_RL = 2, kENCODING_DIFF = 3, kENCODING_DICT = 4,
                    kENCODING_SPARSE = 5, kENCODING_GEOINT = 6, kENCODING_DATE_IN_DAYS = 7 };
enum SQLOps { kEQ = 0, kBW_EQ, kNE, kLT, kGT, kLE, kGE, kAND, kOR, kNOT, kMINUS, kPLUS, kMULTIPLY, kDIVIDE, kMODULO, kUMINUS, kISNULL, kISNOTNULL, kEXISTS, kCAST };
enum SQLAgg { kAVG = 0, kMIN, kMAX, kSUM, kCOUNT, kAPPROX_COUNT_DISTINCT, kAPPROX_QUANTILE, kSAMPLE, kSINGLE_VALUE, kMODE,
              kCOUNT_IF, kSUM_IF, kINVALID_AGG };
enum class JoinType { INNER, LEFT };
enum class ExecutorDeviceType { CPU, GPU };

class SQLTypeInfo {
 public:
  SQLTypeInfo(SQLTypes t = kNULLT, bool notnull = false, EncodingType c = kENCODING_NONE, int comp_param = 0, int dimension = 0)
      : type_(t), notnull_(notnull), comp_(c), comp_param_(comp_param), dimension_(dimension) {}
  SQLTypes get_type() const { return type_; }
  bool get_notnull() const { return notnull_; }
  EncodingType get_compression() const { return comp_; }
  int get_comp_param() const { return comp_param_; }
  bool is_fp() const { return type_ == kFLOAT || type_ == kDOUBLE; }
  // (sqltypes.h: IS_NUMBER / is_decimal / is_string / is_array / is_geometry / is_high_precision_timestamp)
  bool is_decimal() const { return type_ == kDECIMAL || type_ == kNUMERIC; }
  bool is_string() const { return type_ == kTEXT || type_ == kVARCHAR || type_ == kCHAR; }
  bool is_array() const { return type_ == kARRAY; }
  bool is_geometry() const { return type_ == kPOINT; }
  int get_dimension() const { return dimension_; }
  bool is_high_precision_timestamp() const { return type_ == kTIMESTAMP && dimension_ > 0; }
  bool is_boolean() const { return type_ == kBOOLEAN; }
  bool is_time() const { return type_ == kTIME || type_ == kTIMESTAMP || type_ == kDATE; }  // (sqltypes.h is_datetime)
  bool is_integer() const { return type_ == kTINYINT || type_ == kSMALLINT || type_ == kINT || type_ == kBIGINT; }
  // bytes of the SQL type
  int get_logical_size() const {
    switch (type_) {
      case kTINYINT: case kBOOLEAN: return 1;
      case kSMALLINT: return 2;
      case kINT: case kFLOAT: case kTEXT: return 4;
      default: return 8;
    }
  }
  // bytes of one element of the chunk as stored (kENCODING_FIXED / _DICT / _DATE_IN_DAYS: comp_param bits)
  int get_size() const {
    if ((comp_ == kENCODING_FIXED || comp_ == kENCODING_DATE_IN_DAYS || comp_ == kENCODING_DICT) && comp_param_ > 0) return comp_param_ / 8;
    return get_logical_size();
  }

 private:
  SQLTypes type_;
  bool notnull_;
  EncodingType comp_;
  int comp_param_;
  int dimension_;
};

union Datum {
  int8_t tinyintval;
  int16_t smallintval;
  int32_t intval;
  int64_t bigintval;
  float floatval;
  double doubleval;
};

namespace shared {
struct TableKey { int32_t db_id, table_id; };
struct ColumnKey { int32_t db_id, table_id, column_id; };
}  // namespace shared

namespace Analyzer {
class Expr {
 public:
  explicit Expr(const SQLTypeInfo& ti) : type_info(ti) {}
  virtual ~Expr() = default;
  const SQLTypeInfo& get_type_info() const { return type_info; }

 protected:
  SQLTypeInfo type_info;
};
class ColumnVar : public Expr {
 public:
  ColumnVar(const SQLTypeInfo& ti, const shared::ColumnKey& key, int rte_idx) : Expr(ti), key_(key), rte_idx_(rte_idx) {}
  const shared::ColumnKey& getColumnKey() const { return key_; }
  int32_t get_rte_idx() const { return rte_idx_; }  // 0 = outer table, 1 = inner

 private:
  shared::ColumnKey key_;
  int rte_idx_;
};
class Constant : public Expr {
 public:
  Constant(const SQLTypeInfo& ti, bool is_null, Datum v) : Expr(ti), is_null_(is_null), constval_(v) {}
  bool get_is_null() const { return is_null_; }
  Datum get_constval() const { return constval_; }

 private:
  bool is_null_;
  Datum constval_;
};
class UOper : public Expr {
 public:
  UOper(const SQLTypeInfo& ti, SQLOps o, std::shared_ptr<Expr> operand) : Expr(ti), op_(o), operand_(std::move(operand)) {}
  SQLOps get_optype() const { return op_; }
  const Expr* get_operand() const { return operand_.get(); }

 private:
  SQLOps op_;
  std::shared_ptr<Expr> operand_;
};
class BinOper : public Expr {
 public:
  BinOper(const SQLTypeInfo& ti, SQLOps o, std::shared_ptr<Expr> l, std::shared_ptr<Expr> r)
      : Expr(ti), op_(o), left_(std::move(l)), right_(std::move(r)) {}
  SQLOps get_optype() const { return op_; }
  const Expr* get_left_operand() const { return left_.get(); }
  const Expr* get_right_operand() const { return right_.get(); }

 private:
  SQLOps op_;
  std::shared_ptr<Expr> left_, right_;
};
class CaseExpr : public Expr {  // Analyzer.h:1447
 public:
  CaseExpr(const SQLTypeInfo& ti, bool /*has_agg*/,
           const std::list<std::pair<std::shared_ptr<Analyzer::Expr>, std::shared_ptr<Analyzer::Expr>>>& w,
           std::shared_ptr<Analyzer::Expr> e)
      : Expr(ti), expr_pair_list(w), else_expr(std::move(e)) {}
  const std::list<std::pair<std::shared_ptr<Analyzer::Expr>, std::shared_ptr<Analyzer::Expr>>>& get_expr_pair_list() const {
    return expr_pair_list;
  }
  const Expr* get_else_expr() const { return else_expr.get(); }

 private:
  std::list<std::pair<std::shared_ptr<Analyzer::Expr>, std::shared_ptr<Analyzer::Expr>>> expr_pair_list;
  std::shared_ptr<Analyzer::Expr> else_expr;
};
class InValues : public Expr {  // Analyzer.h:641
 public:
  InValues(std::shared_ptr<Analyzer::Expr> a, const std::list<std::shared_ptr<Analyzer::Expr>>& l)
      : Expr(SQLTypeInfo(kBOOLEAN, a->get_type_info().get_notnull())), arg(std::move(a)), value_list(l) {}
  const Expr* get_arg() const { return arg.get(); }
  const std::list<std::shared_ptr<Analyzer::Expr>>& get_value_list() const { return value_list; }

 private:
  std::shared_ptr<Analyzer::Expr> arg;
  std::list<std::shared_ptr<Analyzer::Expr>> value_list;
};
class AggExpr : public Expr {
 public:
  AggExpr(const SQLTypeInfo& ti, SQLAgg a, std::shared_ptr<Expr> arg, bool distinct = false,
          std::shared_ptr<Expr> arg1 = nullptr)
      : Expr(ti), agg_(a), arg_(std::move(arg)), distinct_(distinct), arg1_(std::move(arg1)) {}
  SQLAgg get_aggtype() const { return agg_; }
  Expr* get_arg() const { return arg_.get(); }  // nullptr: COUNT(*); COUNT_IF: the condition
  bool get_is_distinct() const { return distinct_; }
  std::shared_ptr<Expr> get_arg1() const { return arg1_; }  // SUM_IF: the condition (Analyzer.h:1055)

 private:
  SQLAgg agg_;
  std::shared_ptr<Expr> arg_;
  bool distinct_;
  std::shared_ptr<Expr> arg1_;
};
}  // namespace Analyzer

class InputDescriptor {
 public:
  InputDescriptor(int32_t db_id, int32_t table_id, int nest_level) : key_{db_id, table_id}, nest_level_(nest_level) {}
  const shared::TableKey& getTableKey() const { return key_; }
  int getNestLevel() const { return nest_level_; }

 private:
  shared::TableKey key_;
  int nest_level_;
};
class InputColDescriptor {
 public:
  InputColDescriptor(int32_t col_id, int32_t table_id, int32_t db_id, int32_t nest_level)
      : col_id_(col_id), input_desc_(db_id, table_id, nest_level) {}
  int getColId() const { return col_id_; }
  const InputDescriptor& getScanDesc() const { return input_desc_; }

 private:
  int col_id_;
  InputDescriptor input_desc_;
};

struct JoinCondition {
  std::list<std::shared_ptr<Analyzer::Expr>> quals;
  JoinType type;
};
using JoinQualsPerNestingLevel = std::vector<JoinCondition>;

struct RelAlgExecutionUnit {
  std::vector<InputDescriptor> input_descs;
  std::list<std::shared_ptr<const InputColDescriptor>> input_col_descs;
  std::list<std::shared_ptr<Analyzer::Expr>> simple_quals;
  std::list<std::shared_ptr<Analyzer::Expr>> quals;
  JoinQualsPerNestingLevel join_quals;
  std::list<std::shared_ptr<Analyzer::Expr>> groupby_exprs;  // {nullptr}: non-grouped
  std::vector<Analyzer::Expr*> target_exprs;
  size_t scan_limit{0};  // (RelAlgExecutionUnit.h:178) LIMIT + OFFSET of a projection without ORDER BY
};

struct FetchResultFragmentInfo {
  std::vector<std::vector<int64_t>> num_rows;
  std::vector<std::vector<uint64_t>> frag_offsets;
  std::vector<std::vector<int32_t>> frag_ids;
};
struct FetchResult {
  std::vector<std::vector<const int8_t*>> col_buffers;  // [frag][col]: GPU_LEVEL chunk pointers
  FetchResultFragmentInfo fragment_info;
};

enum class ExpressionRangeType { Invalid, Integer, Float, Double };
class ExpressionRange {
 public:
  static ExpressionRange makeIntRange(int64_t lo, int64_t hi, int64_t bucket, bool has_nulls) {
    ExpressionRange r;
    r.type_ = ExpressionRangeType::Integer;
    r.int_min_ = lo; r.int_max_ = hi; r.bucket_ = bucket; r.has_nulls_ = has_nulls;
    return r;
  }
  static ExpressionRange makeDoubleRange(double lo, double hi, bool has_nulls) {
    ExpressionRange r;
    r.type_ = ExpressionRangeType::Double;
    r.fp_min_ = lo; r.fp_max_ = hi; r.has_nulls_ = has_nulls;
    return r;
  }
  static ExpressionRange makeInvalidRange() { return ExpressionRange(); }
  ExpressionRangeType getType() const { return type_; }
  bool hasNulls() const { return has_nulls_; }
  int64_t getIntMin() const { return int_min_; }
  int64_t getIntMax() const { return int_max_; }
  double getFpMin() const { return fp_min_; }
  double getFpMax() const { return fp_max_; }
  int64_t getBucket() const { return bucket_; }

 private:
  ExpressionRangeType type_ = ExpressionRangeType::Invalid;
  bool has_nulls_ = false;
  int64_t int_min_ = 0, int_max_ = 0, bucket_ = 0;
  double fp_min_ = 0, fp_max_ = 0;
};

struct InputTableInfo {
  shared::TableKey table_key;
  int64_t num_tuples;  // info.getNumTuples()
  std::vector<int64_t> fragment_rows;  // info.fragments[i].getNumTuples() (Fragmenter_Namespace::TableInfo); may be empty here
};

class Executor;
// QueryEngine/ExpressionRange.h: the bounds chunk metadata gives an expression.  The mock's Executor carries a
// per-column table the test driver fills; casts / + - * combine bounds like ExpressionRange.cpp does.
ExpressionRange getExpressionRange(const Analyzer::Expr* expr, const std::vector<InputTableInfo>& query_infos,
                                   const Executor* executor);

// accessors of QueryEngine/Descriptors/QueryMemoryDescriptor.h the binding reads for its layout self-check
class QueryMemoryDescriptor {
 public:
  size_t entry_count = 0, row_size = 0, buffer_bytes = 0;
  int8_t compact_byte_width = 8;
  bool keyless = false, columnar = false;
  size_t getEntryCount() const { return entry_count; }
  size_t getRowSize() const { return row_size; }
  int8_t getCompactByteWidth() const { return compact_byte_width; }
  bool hasKeylessHash() const { return keyless; }
  bool didOutputColumnar() const { return columnar; }
  size_t getBufferSizeBytes(ExecutorDeviceType) const { return buffer_bytes; }
};

class RowSetMemoryOwner {};
class ResultSetStorage {
 public:
  explicit ResultSetStorage(size_t bytes) : buf_(bytes) {}
  int8_t* getUnderlyingBuffer() { return buf_.data(); }
  size_t bytes() const { return buf_.size(); }

 private:
  std::vector<int8_t> buf_;
};
struct TargetInfo {
  bool is_agg;
  SQLAgg agg_kind;
  SQLTypeInfo sql_type, agg_arg_type;
  bool skip_null_val, is_distinct;
};
class ResultSet {
 public:
  ResultSet(std::vector<TargetInfo> targets, ExecutorDeviceType, const QueryMemoryDescriptor& qmd,
            std::shared_ptr<RowSetMemoryOwner>, unsigned, unsigned)
      : targets_(std::move(targets)), qmd_(qmd) {}
  ResultSetStorage* allocateStorage() {
    storage_ = std::make_unique<ResultSetStorage>(qmd_.getBufferSizeBytes(ExecutorDeviceType::CPU));
    return storage_.get();
  }
  const ResultSetStorage* getStorage() const { return storage_.get(); }
  const QueryMemoryDescriptor& getQueryMemDesc() const { return qmd_; }

 private:
  std::vector<TargetInfo> targets_;
  QueryMemoryDescriptor qmd_;
  std::unique_ptr<ResultSetStorage> storage_;
};
using ResultSetPtr = std::shared_ptr<ResultSet>;

class QueryExecutionError : public std::runtime_error {
 public:
  explicit QueryExecutionError(int32_t code) : std::runtime_error("Query execution failed with error code " + std::to_string(code)), code_(code) {}
  int32_t getErrorCode() const { return code_; }

 private:
  int32_t code_;
};

class Executor {
 public:
  void* getCudaStream(int /*device_id*/) const { return nullptr; }  // the library's own stream
  std::shared_ptr<RowSetMemoryOwner> getRowSetMemoryOwner() const { return owner_; }
  // mock catalog + chunk metadata: (table_id, column_id) -> type and range
  std::map<std::pair<int, int>, SQLTypeInfo> column_types;
  std::map<std::pair<int, int>, ExpressionRange> column_ranges;

 private:
  std::shared_ptr<RowSetMemoryOwner> owner_ = std::make_shared<RowSetMemoryOwner>();
};
// Catalog lookup of a column's SQLTypeInfo (get_column_descriptor(...)->columnType in a HeavyDB build)
inline SQLTypeInfo get_column_type(int col_id, const shared::TableKey& tk, const Executor* ex) {
  return ex->column_types.at({tk.table_id, col_id});
}

extern bool g_bigint_count;
#define CHECK_EQ(a, b) do { if (!((a) == (b))) throw std::logic_error(std::string("CHECK_EQ failed: ") + #a + " == " + #b); } while (0)
#define CHECK(a) do { if (!(a)) throw std::logic_error(std::string("CHECK failed: ") + #a); } while (0)
