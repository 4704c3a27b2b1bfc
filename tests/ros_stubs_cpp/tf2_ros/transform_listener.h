#pragma once
#include <tf2_ros/buffer.h>
namespace tf2_ros {
class TransformListener {
 public:
  explicit TransformListener(Buffer&) {}
};
}  // namespace tf2_ros
