#pragma once
#include <geometry_msgs/Pose.h>
namespace geometry_msgs {
struct PoseArray {
  typedef std::shared_ptr<const PoseArray> ConstPtr;
  std_msgs::Header header;
  std::vector<Pose> poses;
};
}  // namespace geometry_msgs
