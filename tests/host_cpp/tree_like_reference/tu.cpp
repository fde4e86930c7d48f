// One translation unit that uses the planners' types (Coordinate, Box: only the headers BEHIND the shim
// have planner_only_helper) together with slam::MCL, under the include order INTEGRATION.md §A documents.
#include "mcl.h"
#include "raycast.h"
#include "fake_lidar.h"
#include "motion.h"
#include "util.h"
#if !defined(TREE_POSE_H_SEEN) || !defined(TREE_UTIL_H_SEEN)
#error "the shim headers shadowed the headers behind them"
#endif
int tu_uses_everything(slam::MCL* mcl)
{
    slam::Coordinate c{1, 2};
    slam::Box b{0, 0, 3, 4};
    slam::Particle p{};
    (void)static_cast<slam::Pose (*)(slam::Particle&, double, double, double)>(&slam::raycast_mapping);
    (void)static_cast<double (*)(double, double, slam::Particle&, double, double)>(&slam::measurement_model_beam);
    return slam::planner_only_helper(c) + b.stop_i + (mcl ? (int)mcl->get_particles().size() : 0) + (int)p.weight;
}
