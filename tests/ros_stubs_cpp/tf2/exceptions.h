#pragma once
#include <stdexcept>
namespace tf2 {
class TransformException : public std::runtime_error {
 public:
  explicit TransformException(const std::string& m) : std::runtime_error(m) {}
};
class LookupException : public TransformException {
 public:
  explicit LookupException(const std::string& m) : TransformException(m) {}
};
}  // namespace tf2
