// TEMPORARY: entry points not implemented yet (replaced by mapping.hip / odometry.hip / factors.hip).
#include "common.hpp"
using namespace scal;
#define NI(name) set_error(#name ": not implemented yet"); return SCAL_E_STATE;
extern "C" {
int scal_odom_create(const scal_odom_config*, scal_odom_t**) { NI(scal_odom_create) }
void scal_odom_destroy(scal_odom_t*) {}
int scal_odom_step(scal_odom_t*, const float*, int, const float*, int, const float*, int, const float*, int, double*, double*, double*, double*, scal_odom_stats*) { NI(scal_odom_step) }
int scal_odom_step_features(scal_odom_t*, scal_features_t*, double*, double*, double*, double*, scal_odom_stats*) { NI(scal_odom_step_features) }
int scal_factors_eval(int, int, const int*, const double*, const double*, const double*, const double*, double*, double*, double*) { NI(scal_factors_eval) }
}
