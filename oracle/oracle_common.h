/* oracle/oracle_common.h — TEST INFRASTRUCTURE ONLY.
 *
 * Shared plain-C types for the CPU oracles.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load anything under oracle/.  The product
 * path (hyperpose_amd/, include/) never includes, links or calls this.
 *
 * Layout of o_human mirrors hyperpose::human_t
 * (reference include/hyperpose/utility/human.hpp:14-31): 18 x {bool,f32 x,f32 y,f32 score} + f32 score.
 */
#ifndef ORACLE_COMMON_H
#define ORACLE_COMMON_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O_COCO_N_PARTS 18 /* human.hpp:10 */
#define O_COCO_N_PAIRS 19 /* human.hpp:11 */

typedef struct {
    int32_t has_value;
    float x, y, score;
} o_body_part;

typedef struct {
    o_body_part parts[O_COCO_N_PARTS];
    float score;
} o_human;

/* peak_info of src/post_process.hpp:126-131 */
typedef struct {
    int32_t part_id;
    int32_t x, y;
    float score;
    int32_t id;
} o_peak;

/* connection of src/paf.cpp:7-13 (cid == peak_id in the reference) + the limb it belongs to */
typedef struct {
    int32_t pair_id;
    int32_t cid1, cid2;
    float score;
} o_conn;

#ifdef __cplusplus
}
#endif
#endif
