#pragma once
#include <memory>
namespace std_msgs {
struct Bool {
  typedef std::shared_ptr<const Bool> ConstPtr;
  bool data = false;
};
}  // namespace std_msgs
