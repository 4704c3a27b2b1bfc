#pragma once
#include <geometry_msgs/Pose.h>
namespace geometry_msgs {
struct TransformStamped {
  std_msgs::Header header;
  std::string child_frame_id;
  Transform transform;
};
}  // namespace geometry_msgs
