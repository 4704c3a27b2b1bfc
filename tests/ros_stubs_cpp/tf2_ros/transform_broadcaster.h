#pragma once
#include <tf2_ros/buffer.h>
namespace tf2_ros {
class TransformBroadcaster {
 public:
  void sendTransform(const geometry_msgs::TransformStamped& t) { stub::broadcasts().push_back(t); }
};
}  // namespace tf2_ros
