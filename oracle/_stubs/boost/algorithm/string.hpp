// TEST INFRASTRUCTURE (oracle/Makefile `ref`): Boost is not in this image.  QueryEngine/CalciteDeserializerUtils.cpp
// uses boost::iequals in its date-part name parsers, which this build never calls (only get_agg_type is used).
#pragma once
#include <cctype>
#include <string>
namespace boost {
inline bool iequals(const std::string& a, const std::string& b) {
  if (a.size() != b.size()) {
    return false;
  }
  for (size_t i = 0; i < a.size(); ++i) {
    if (std::tolower(static_cast<unsigned char>(a[i])) != std::tolower(static_cast<unsigned char>(b[i]))) {
      return false;
    }
  }
  return true;
}
}  // namespace boost
