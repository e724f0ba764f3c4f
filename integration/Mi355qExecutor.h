// Mi355qExecutor.h — the HeavyDB-side binding of libmi355q.so (what would live in QueryEngine/).
#pragma once

#include <string>

#include "mi355q.h"

#ifdef MI355Q_GLUE_MOCK_HEADERS
#include "mock/heavydb_mock.h"   // this repository: stand-ins for the HeavyDB headers (see that file)
#else
#include "QueryEngine/ColumnFetcher.h"
#include "QueryEngine/Descriptors/QueryMemoryDescriptor.h"
#include "QueryEngine/ErrorHandling.h"
#include "QueryEngine/Execute.h"
#include "QueryEngine/ExpressionRange.h"
#include "QueryEngine/RelAlgExecutionUnit.h"
#include "QueryEngine/ResultSet.h"
#endif

#include "Mi355qTranslate.h"

namespace mi355q_glue {

// RelAlgExecutionUnit (RelAlgExecutionUnit.h:167-218) + chunk metadata -> mi355q_plan.  Throws
// std::runtime_error for a step the plan ABI cannot state (the caller then keeps the native path, the way
// ExecutorType::Extern falls back on NativeExecutionError, RelAlgExecutor.cpp:1430-1449).
mi355q_plan to_plan(const RelAlgExecutionUnit& ra, const std::vector<InputTableInfo>& query_infos,
                    const Executor* executor, const mi355q_join_table* join_table,
                    size_t max_groups_buffer_entry_guess, bool output_columnar_hint);

// Called from ExecutionKernel::runImpl in place of the JIT launch, after fetchChunks (ExecutionKernel.cpp:270-292;
// the seam run_query_external uses, ExternalExecutor.cpp:517-530).
ResultSetPtr run_query_mi355q(const RelAlgExecutionUnit& ra, const FetchResult& fetch_result,
                              const std::vector<InputTableInfo>& query_infos,
                              const QueryMemoryDescriptor& query_mem_desc, Executor* executor, int device_id,
                              size_t max_groups_buffer_entry_guess, const mi355q_join_table* join_table,
                              const std::vector<const int8_t*>& inner_col_buffers, int64_t inner_num_rows);

// ExecutionOptions::just_explain (Execute.cpp: executeWorkUnit returns the generated IR as an "explanation" ResultSet,
// RelAlgExecutor.cpp:executeRelAlgQuery): a fixed family has no IR to show, so the explanation of a step is the ROUTE —
// the derived-plan stages and the kernel family mi355q_execute would take for this plan over fragments of these sizes
// (mi355q_explain: nothing is launched, nothing allocated).
std::string explain_query_mi355q(const RelAlgExecutionUnit& ra, const std::vector<InputTableInfo>& query_infos,
                                 const Executor* executor, int device_id, size_t max_groups_buffer_entry_guess,
                                 const mi355q_join_table* join_table, int64_t inner_num_rows, bool output_columnar_hint);

}  // namespace mi355q_glue
