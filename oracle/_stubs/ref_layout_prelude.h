// TEST INFRASTRUCTURE (oracle/Makefile `ref`): force-included ahead of the reference's sources when they are
// compiled with -DNO_BOOST.  Shared/sqldefs.h hides its enum toString() overloads under NO_BOOST while
// Shared/TargetInfo.h still calls them from TargetInfo::toString(); these stand in (never called by the shim).
#pragma once
#include <string>
#include <unordered_map>  // QueryEngine/CalciteDeserializerUtils.h uses it without including it
#include "Shared/sqldefs.h"
inline std::string toString(const SQLAgg& kind) { return std::to_string((int)kind); }
inline std::string toString(const SQLOps& op) { return std::to_string((int)op); }
inline std::string toString(const JoinType& t) { return std::to_string((int)t); }
