// analyzer_link_stubs.cpp — link-line glue for integration/real/real_headers_check.cpp ONLY (test infrastructure).
// Analyzer/Analyzer.cpp cannot be compiled in this image (it includes Calcite/Calcite.h -> gen-cpp/calciteserver_types.h,
// Thrift-generated and absent), so the out-of-line virtuals the Analyzer vtables name are aborting stand-ins: the check
// only constructs expression objects and reads them through the inline accessors the binding uses.  None of this
// computes a value under test.  (oracle/ref_layout_shim.cpp carries the same glue for its own link.)
#include <cstdlib>

#include "Analyzer/Analyzer.h"

bool g_bigint_count{false};
std::string SQLTypeInfo::type_name[kSQLTYPE_LAST];  // Shared/Datum.cpp:42 (Datum.cpp needs Boost)
std::string SQLTypeInfo::comp_name[kENCODING_LAST];

#define REF_STUB \
  { abort(); }

namespace Analyzer {
std::shared_ptr<Expr> Expr::add_cast(const SQLTypeInfo&) REF_STUB
size_t Expr::get_num_column_vars(const bool) const REF_STUB
void Expr::add_unique(std::list<const Expr*>&) const REF_STUB

void ColumnVar::check_group_by(const std::list<std::shared_ptr<Expr>>&) const REF_STUB
std::shared_ptr<Expr> ColumnVar::deep_copy() const REF_STUB
void ColumnVar::group_predicates(std::list<const Expr*>&, std::list<const Expr*>&, std::list<const Expr*>&) const REF_STUB
std::shared_ptr<Expr> ColumnVar::rewrite_with_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> ColumnVar::rewrite_with_child_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> ColumnVar::rewrite_agg_to_var(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::string ColumnVar::toString() const REF_STUB
bool ColumnVar::operator==(const Expr&) const REF_STUB

Constant::~Constant() {}
void Constant::set_null_value() {}  // (called by the constructor of a NULL constant; the binding never reads its value)
std::shared_ptr<Expr> Constant::deep_copy() const REF_STUB
std::shared_ptr<Expr> Constant::add_cast(const SQLTypeInfo&) REF_STUB
bool Constant::operator==(const Expr&) const REF_STUB
std::string Constant::toString() const REF_STUB

void UOper::check_group_by(const std::list<std::shared_ptr<Expr>>&) const REF_STUB
std::shared_ptr<Expr> UOper::deep_copy() const REF_STUB
void UOper::group_predicates(std::list<const Expr*>&, std::list<const Expr*>&, std::list<const Expr*>&) const REF_STUB
bool UOper::operator==(const Expr&) const REF_STUB
std::string UOper::toString() const REF_STUB
void UOper::find_expr(std::function<bool(const Expr*)>, std::list<const Expr*>&) const REF_STUB
std::shared_ptr<Expr> UOper::add_cast(const SQLTypeInfo&) REF_STUB

void BinOper::check_group_by(const std::list<std::shared_ptr<Expr>>&) const REF_STUB
std::shared_ptr<Expr> BinOper::deep_copy() const REF_STUB
std::shared_ptr<Expr> BinOper::normalize_simple_predicate(int&) const REF_STUB
void BinOper::group_predicates(std::list<const Expr*>&, std::list<const Expr*>&, std::list<const Expr*>&) const REF_STUB
bool BinOper::operator==(const Expr&) const REF_STUB
std::string BinOper::toString() const REF_STUB
void BinOper::find_expr(std::function<bool(const Expr*)>, std::list<const Expr*>&) const REF_STUB

std::shared_ptr<Expr> CaseExpr::deep_copy() const REF_STUB
void CaseExpr::check_group_by(const std::list<std::shared_ptr<Expr>>&) const REF_STUB
void CaseExpr::group_predicates(std::list<const Expr*>&, std::list<const Expr*>&, std::list<const Expr*>&) const REF_STUB
void CaseExpr::collect_rte_idx(std::set<int>&) const REF_STUB
void CaseExpr::collect_column_var(std::set<const ColumnVar*, bool (*)(const ColumnVar*, const ColumnVar*)>&, bool) const REF_STUB
std::shared_ptr<Expr> CaseExpr::rewrite_with_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> CaseExpr::rewrite_with_child_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> CaseExpr::rewrite_agg_to_var(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
bool CaseExpr::operator==(const Expr&) const REF_STUB
std::string CaseExpr::toString() const REF_STUB
void CaseExpr::find_expr(std::function<bool(const Expr*)>, std::list<const Expr*>&) const REF_STUB
std::shared_ptr<Expr> CaseExpr::add_cast(const SQLTypeInfo&) REF_STUB
void CaseExpr::get_domain(DomainSet&) const REF_STUB

// (the constructor is out of line in Analyzer.cpp:1707-1709, where the type's nullability also looks at the list)
InValues::InValues(std::shared_ptr<Analyzer::Expr> a, const std::list<std::shared_ptr<Analyzer::Expr>>& l)
    : Expr(kBOOLEAN, a->get_type_info().get_notnull()), arg(a), value_list(l) {}
std::shared_ptr<Expr> InValues::deep_copy() const REF_STUB
void InValues::group_predicates(std::list<const Expr*>&, std::list<const Expr*>&, std::list<const Expr*>&) const REF_STUB
std::shared_ptr<Expr> InValues::rewrite_with_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> InValues::rewrite_with_child_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> InValues::rewrite_agg_to_var(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
bool InValues::operator==(const Expr&) const REF_STUB
std::string InValues::toString() const REF_STUB
void InValues::find_expr(std::function<bool(const Expr*)>, std::list<const Expr*>&) const REF_STUB

std::shared_ptr<Expr> AggExpr::deep_copy() const REF_STUB
void AggExpr::group_predicates(std::list<const Expr*>&, std::list<const Expr*>&, std::list<const Expr*>&) const REF_STUB
std::shared_ptr<Expr> AggExpr::rewrite_with_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> AggExpr::rewrite_with_child_targetlist(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
std::shared_ptr<Expr> AggExpr::rewrite_agg_to_var(const std::vector<std::shared_ptr<TargetEntry>>&) const REF_STUB
bool AggExpr::operator==(const Expr&) const REF_STUB
std::string AggExpr::toString() const REF_STUB
void AggExpr::find_expr(std::function<bool(const Expr*)>, std::list<const Expr*>&) const REF_STUB
}  // namespace Analyzer
