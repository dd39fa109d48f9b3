// Mirror of scripts/run-msm-ed-377.ts:  node js/scripts/run-msm-ed-377.mjs <n> [gpus] [--evaluate] [--json]
import { edOnBls12377Params } from "../concrete/params.mjs";
import { main } from "./msm-drivers.mjs";
main(edOnBls12377Params, "te").catch((e) => { console.error(e); process.exit(1); });
