// Stand-in for <ros/ros.h> (tests/ros_stubs_cpp/README.md): an in-process "master".
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

namespace ros {

struct Duration {
  double d = 0.0;
  Duration() {}
  explicit Duration(double s) : d(s) {}
  double toSec() const { return d; }
};
struct Time {
  double t = 0.0;
  Time() {}
  explicit Time(double s) : t(s) {}
  double toSec() const { return t; }
  static Time now();
};
struct TimerEvent {};

namespace stub {
struct Master {
  std::map<std::string, std::string> params;                               // private parameters of the node
  std::map<std::string, std::function<void(const void*)>> subs;            // topic -> callback(shared_ptr<const M>*)
  std::map<std::string, std::vector<std::shared_ptr<void>>> sent;          // topic -> published messages
  std::map<std::string, uint32_t> pub_queue;
  std::vector<std::pair<double, std::function<void(const TimerEvent&)>>> timers;
  std::vector<std::string> log;                                            // "I ...", "W ...", "E ..."
  double now = 0.0;
  int spinner_threads = 0;
  std::string node_name;
  std::mutex mu;   // publishing and logging are thread-safe in roscpp: so are they here (tests/host_san: four threads)
};
inline Master& master() {
  static Master m;
  return m;
}
template <class M>
void deliver(const std::string& topic, const M& msg) {
  std::shared_ptr<const M> p = std::make_shared<M>(msg);
  master().subs.at(topic)(&p);
}
inline void logf(char level, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  std::lock_guard<std::mutex> lk(master().mu);
  master().log.push_back(std::string(1, level) + " " + buf);
  std::fprintf(stderr, "[%c] %s\n", level, buf);
}
}  // namespace stub

inline Time Time::now() { return Time(stub::master().now); }

class Publisher {
 public:
  Publisher() {}
  explicit Publisher(const std::string& t) : topic_(t) {}
  template <class M>
  void publish(const M& m) const {
    std::lock_guard<std::mutex> lk(stub::master().mu);
    stub::master().sent[topic_].push_back(std::make_shared<M>(m));
  }

 private:
  std::string topic_;
};
class Subscriber {};
class Timer {};

class NodeHandle {
 public:
  explicit NodeHandle(const std::string& ns = std::string()) : ns_(ns) {}
  template <class T>
  bool param(const std::string& name, T& var, const T& def) const {
    const auto& p = stub::master().params;
    const auto it = p.find(name);
    if (it == p.end()) {
      var = def;
      return false;
    }
    parse(it->second, var);
    return true;
  }
  template <class M>
  Publisher advertise(const std::string& topic, uint32_t queue) {
    stub::master().sent[topic];
    stub::master().pub_queue[topic] = queue;
    return Publisher(topic);
  }
  template <class M, class T>
  Subscriber subscribe(const std::string& topic, uint32_t, void (T::*fp)(const std::shared_ptr<const M>&), T* obj) {
    stub::master().subs[topic] = [fp, obj](const void* m) { (obj->*fp)(*static_cast<const std::shared_ptr<const M>*>(m)); };
    return Subscriber();
  }
  template <class T>
  Timer createTimer(Duration period, void (T::*fp)(const TimerEvent&), T* obj) {
    stub::master().timers.push_back({period.toSec(), [fp, obj](const TimerEvent& e) { (obj->*fp)(e); }});
    return Timer();
  }

 private:
  static void parse(const std::string& s, std::string& v) { v = s; }
  static void parse(const std::string& s, int& v) { v = std::stoi(s); }
  static void parse(const std::string& s, double& v) { v = std::stod(s); }
  static void parse(const std::string& s, bool& v) { v = s == "true" || s == "1"; }
  std::string ns_;
};

class AsyncSpinner {
 public:
  explicit AsyncSpinner(int threads) { stub::master().spinner_threads = threads; }
  void start() {}
};

void init(int& argc, char** argv, const std::string& name);   // scenario.cpp
void waitForShutdown();                                       // scenario.cpp: plays the messages

}  // namespace ros

#define ROS_INFO(...) ::ros::stub::logf('I', __VA_ARGS__)
#define ROS_WARN(...) ::ros::stub::logf('W', __VA_ARGS__)
#define ROS_ERROR(...) ::ros::stub::logf('E', __VA_ARGS__)
#define ROS_WARN_THROTTLE(period, ...) ::ros::stub::logf('W', __VA_ARGS__)
